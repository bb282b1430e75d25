"""Mirror of reference backbones/resnet_ppm.py:9-13."""
import torch.nn as nn

from .resnet import resnet50
from .resnet_dilated import ResnetDilated
from .ppm import PPMDeepsup


def resnet50dilated_ppm(resnet_pretrained=False, **kwargs):
    resnet = resnet50(pretrained=resnet_pretrained)
    resnet_dilated = ResnetDilated(resnet, dilate_scale=8)
    ppm = PPMDeepsup(**kwargs)
    return nn.Sequential(resnet_dilated, ppm)
