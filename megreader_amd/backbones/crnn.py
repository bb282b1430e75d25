"""CRNN convolutional backbone on HIP kernels.

Mirror of reference backbones/crnn.py:4-63 -- same constructor, same nn.Sequential nesting (hence the same
``state_dict`` keys: ``cnn.{i}.0.0.weight`` ...), same quirk that BatchNorm layers carry no activation
(crnn.py:46-55).  ReLU is fused into the convolution epilogue; the FusedReLU module only
keeps the Sequential indices of the reference.
"""
import torch.nn as nn

from ..nn import Conv2d, BatchNorm2d, MaxPool2d, FusedReLU
from ..nn import functional as F


class _Stem(nn.Sequential):
    """conv0 + relu0 + pooling0 (reference crnn.py:17-19).  Same children / state_dict keys as the plain
    Sequential; when the input is the raw image (no gradient wanted) the three ops run as one fused kernel each way
    (megreader_amd/csrc/stem.hip), otherwise the generic Conv2d -> MaxPool2d path runs."""

    def forward(self, input):
        conv, pool = self[0][0], self[1]
        if F.stem_eligible(input, conv.weight, conv.stride, conv.padding, conv.dilation, _pair(pool.kernel_size),
                           _pair(pool.stride), _pair(pool.padding)):
            return F.stem_conv_relu_pool(input, conv.weight, conv.bias)
        return super().forward(input)


class _ConvReluPool(nn.Sequential):
    """conv + ReLU + max-pool (reference crnn.py:20-22, 26-28, 32-34).  Same children / state_dict keys as the plain Sequential;
    in bf16, where the convolution's tiles can be cut on pooling-window boundaries, the three ops are ONE forward launch whose
    epilogue pools the tile out of LDS (megreader_amd/csrc/igemm_core.h: EpiPool) -- the full-resolution activation (67 MB for
    conv1 at N = 256) is neither written nor re-read; backward is the pool's and the convolution's, unchanged."""

    def forward(self, input):
        conv, pool = self[0][0], self[1]
        if F.conv_relu_pool_eligible(input, conv.weight, conv.stride, conv.padding, conv.dilation, _pair(pool.kernel_size),
                                     _pair(pool.stride), _pair(pool.padding)):
            return F.conv_relu_pool(input, conv.weight, conv.bias, conv.padding, _pair(pool.kernel_size), _pair(pool.stride),
                                    _pair(pool.padding))
        return super().forward(input)


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class CRNN(nn.Module):

    def __init__(self, imgH, nc, nclass, nh):
        super(CRNN, self).__init__()
        assert imgH % 16 == 0, 'imgH has to be a multiple of 16'

        self.kernels = [3, 3, 3, 3, 3, 3, 2]
        self.paddings = [1, 1, 1, 1, 1, 1, 0]
        self.strides = [1, 1, 1, 1, 1, 1, 1]
        self.channels = [64, 128, 256, 256, 512, 512, 512, nc]

        # conv+ReLU stages feed a max-pool: the pool's backward also applies the ReLU mask (one pass saved)
        conv0 = _Stem(self._make_layer(0, pooled=True), MaxPool2d((2, 2), relu_input=True))
        conv1 = _ConvReluPool(self._make_layer(1, pooled=True), MaxPool2d((2, 2), relu_input=True))
        conv2 = self._make_layer(2, True)
        conv3 = _ConvReluPool(self._make_layer(3, pooled=True), MaxPool2d((2, 2), (2, 1), (0, 1), relu_input=True))
        conv4 = self._make_layer(4, True)
        conv5 = _ConvReluPool(self._make_layer(5, pooled=True), MaxPool2d((2, 2), (2, 1), (0, 1), relu_input=True))
        conv6 = self._make_layer(6, True)

        self.cnn = nn.Sequential(conv0, conv1, conv2, conv3, conv4, conv5, conv6)
        # conv3 / conv5 are the only consumers of the BatchNorm outputs of stages 2 / 4 (no activation, no pooling in between,
        # reference crnn.py:46-55): those BatchNorms' backward reductions ride in the dgrad epilogues (nn.Conv2d)
        conv3[0][0].sole_consumer_of_bn = True
        conv5[0][0].sole_consumer_of_bn = True

    def _make_layer(self, i, batch_normalization=False, pooled=False):
        in_channel = self.channels[i - 1]
        out_channel = self.channels[i]
        layer = list()
        layer.append(Conv2d(in_channel, out_channel, self.kernels[i], self.strides[i], self.paddings[i],
                            fuse_relu=not batch_normalization,
                            relu_grad_downstream=pooled and not batch_normalization))
        if batch_normalization:
            layer.append(BatchNorm2d(out_channel))
        else:
            layer.append(FusedReLU())
        return nn.Sequential(*layer)

    def forward(self, input):
        # conv features: logical [N, 512, 1, W/4+1], NHWC strides, compute dtype
        return self.cnn(input)


def crnn_backbone(imgH=32, nc=3, nclass=37, nh=256):
    return CRNN(imgH, nc, nclass, nh)
