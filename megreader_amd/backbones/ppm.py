"""Pyramid pooling module on the HIP layers.  Mirror of reference backbones/ppm.py:6-44 (same module tree incl. the
unused `cbr_deepsup`, which therefore never receives a gradient)."""
import torch.nn as nn

from ..nn import Conv2d, BatchNorm2d, FusedReLU, AdaptiveAvgPool2d, Dropout2d
from ..nn import functional as F
from .base import conv3x3_bn_relu


class PPMDeepsup(nn.Module):
    def __init__(self, inner_channels=256, fc_dim=2048, pool_scales=(1, 2, 3, 6)):
        super(PPMDeepsup, self).__init__()
        self.ppm = []
        for scale in pool_scales:
            self.ppm.append(nn.Sequential(
                AdaptiveAvgPool2d(scale),
                Conv2d(fc_dim, 512, kernel_size=1, bias=False),
                BatchNorm2d(512, fuse_relu=True),
                FusedReLU()))
        self.ppm = nn.ModuleList(self.ppm)
        self.cbr_deepsup = conv3x3_bn_relu(fc_dim // 2, fc_dim // 4, 1)
        self.conv_last = nn.Sequential(
            Conv2d(fc_dim + len(pool_scales) * 512, 512, kernel_size=3, padding=1, bias=False),
            BatchNorm2d(512, fuse_relu=True),
            FusedReLU(),
            Dropout2d(0.1),
            Conv2d(512, inner_channels, kernel_size=1))

    def forward(self, conv_out, segSize=None):
        conv5 = conv_out[-1]
        # reference: pool_scale(conv5) per branch = AdaptiveAvgPool2d(scale) -> conv -> bn -> relu.  The four pools read the
        # same map: one launch (F.adaptive_avg_pool2d_multi), then the rest of each branch
        pooled = F.adaptive_avg_pool2d_multi(conv5, [branch[0].output_size for branch in self.ppm])
        # torch.cat([conv5] + [interpolate(branch, conv5 size, bilinear)], 1): branches resized into their slices of the buffer
        ppm_out = F.cat_bilinear(conv5, [branch[1:](p) for branch, p in zip(self.ppm, pooled)])
        return self.conv_last(ppm_out)
