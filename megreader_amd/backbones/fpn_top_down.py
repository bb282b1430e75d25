"""Mirror of reference backbones/fpn_top_down.py:6-30 on the HIP layers (1x1 lateral convs without bias, bilinear
upsample-add with align_corners=False semantics -- reference quirk Q15 --, 3x3 merge conv)."""
import torch.nn as nn

from ..nn import Conv2d
from ..nn import functional as F


class FPNTopDown(nn.Module):
    def __init__(self, pyramid_channels, feature_channel):
        nn.Module.__init__(self)
        self.reduction_layers = nn.ModuleList()
        for pyramid_channel in pyramid_channels:
            self.reduction_layers.append(
                Conv2d(pyramid_channel, feature_channel, kernel_size=1, stride=1, padding=0, bias=False))
        self.merge_layer = Conv2d(feature_channel, feature_channel, kernel_size=3, stride=1, padding=1, bias=False)

    def upsample_add(self, x, y):
        return F.upsample_add(x, y)

    def forward(self, pyramid_features):
        feature = None
        for pyramid_feature, reduction_layer in zip(pyramid_features, self.reduction_layers):
            pyramid_feature = reduction_layer(pyramid_feature)
            if feature is None:
                feature = pyramid_feature
            else:
                feature = self.upsample_add(feature, pyramid_feature)
        return self.merge_layer(feature)
