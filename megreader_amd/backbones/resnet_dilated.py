"""Mirror of reference backbones/resnet_dilated.py:5-69: stride -> dilation rewrite of layers 3/4 (output stride 8)."""
from functools import partial

import torch.nn as nn


class ResnetDilated(nn.Module):
    def __init__(self, orig_resnet, dilate_scale=8):
        super(ResnetDilated, self).__init__()
        if dilate_scale == 8:
            orig_resnet.layer3.apply(partial(self._nostride_dilate, dilate=2))
            orig_resnet.layer4.apply(partial(self._nostride_dilate, dilate=4))
        elif dilate_scale == 16:
            orig_resnet.layer4.apply(partial(self._nostride_dilate, dilate=2))
        # everything except AvgPool / FC / smooth
        for name in ("conv1", "bn1", "relu1", "conv2", "bn2", "relu2", "conv3", "bn3", "relu3", "maxpool", "layer1",
                     "layer2", "layer3", "layer4"):
            setattr(self, name, getattr(orig_resnet, name))

    def _nostride_dilate(self, m, dilate):
        if m.__class__.__name__.find('Conv') != -1:
            if m.stride == (2, 2):          # the convolution with stride
                m.stride = (1, 1)
                if m.kernel_size == (3, 3):
                    m.dilation = (dilate // 2, dilate // 2)
                    m.padding = (dilate // 2, dilate // 2)
            elif m.kernel_size == (3, 3):   # other convolutions
                m.dilation = (dilate, dilate)
                m.padding = (dilate, dilate)

    def forward(self, x, return_feature_maps=True):
        conv_out = []
        x = self.bn1(self.conv1(x))
        x = self.bn2(self.conv2(x))
        x = self.bn3(self.conv3(x))
        x = self.maxpool(x)
        x = self.layer1(x)
        conv_out.append(x)
        x = self.layer2(x)
        conv_out.append(x)
        x = self.layer3(x)
        conv_out.append(x)
        x = self.layer4(x)
        conv_out.append(x)
        if return_feature_maps:
            return conv_out
        return x
