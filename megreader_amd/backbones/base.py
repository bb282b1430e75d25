"""Mirror of reference backbones/base.py:5-16 on the HIP layers."""
import torch.nn as nn

from ..nn import Conv2d, BatchNorm2d, FusedReLU


def conv3x3(in_planes, out_planes, stride=1, has_bias=False):
    "3x3 convolution with padding"
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=has_bias)


def conv3x3_bn_relu(in_planes, out_planes, stride=1):
    return nn.Sequential(conv3x3(in_planes, out_planes, stride), BatchNorm2d(out_planes, fuse_relu=True), FusedReLU())
