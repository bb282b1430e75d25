"""Deep-stem ResNet (18/34/50/101/152) on the HIP layers.

Mirror of reference backbones/resnet.py:39-333: same module tree (=> same `state_dict` keys and the same RNG
consumption order of the default initialisation, resnet.py:214-221), same forward contract (returns x2..x5), same
quirks: `inplanes = 128` deep stem (:190-201); `smooth` / `fc` / `avgpool` exist but are unused (:210-213);
`stage_with_dcn` is stored but the `dcn` dict is passed to layers 2-4 unconditionally (:203-209, SURVEY B Q13);
`dilation` only reaches the 1x1 downsample conv (:228-235).  BatchNorm+ReLU and the residual add run fused in the
BatchNorm kernel.  Pretrained-weight download (resnet.py:271-309) needs a network: `pretrained=True` raises here.
"""
import math

import torch
import torch.nn as nn

from ..nn import Conv2d, BatchNorm2d, MaxPool2d, FusedReLU, Linear

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152', 'deformable_resnet50']


def constant_init(module, constant, bias=0):
    nn.init.constant_(module.weight, constant)
    if hasattr(module, 'bias'):
        nn.init.constant_(module.bias, bias)


_sync_bn_source = None     # callable -> bool, installed by megreader_amd.dropin.install(): the reference's config.sync_bn


def set_sync_bn_source(fn):
    """Explicit switch for reference resnet.py:26-30 (`if config.sync_bn: apex.parallel.SyncBatchNorm`).  The reference reads
    its own top-level `config` module; here the drop-in layer registers a callable that resolves THAT module (checked
    against the reference root) -- nothing is looked up by name in sys.modules.  None: plain BatchNorm2d."""
    global _sync_bn_source
    _sync_bn_source = fn


def bn(*args, **kwargs):
    # reference resnet.py:26-30: apex.parallel.SyncBatchNorm when config.sync_bn (default False, config.py:14; truthiness)
    if _sync_bn_source is not None and _sync_bn_source():
        from ..apex.parallel import SyncBatchNorm
        return SyncBatchNorm(*args, **kwargs)
    return BatchNorm2d(*args, **kwargs)


def conv3x3(in_planes, out_planes, stride=1):
    """3x3 convolution with padding"""
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _make_dcn(dcn, planes, stride):
    """conv2_offset + deformable conv2 of a block (resnet.py:56-77,125-142)."""
    deformable_groups = dcn.get('deformable_groups', 1)
    if not dcn.get('modulated', False):
        from ..assets.ops.dcn import DeformConv as conv_op
        offset_channels = 18
    else:
        from ..assets.ops.dcn import ModulatedDeformConv as conv_op
        offset_channels = 27
    # NOTE (reference quirk Q10): the offset conv always has stride 1, even when conv2 has stride 2
    conv2_offset = Conv2d(planes, deformable_groups * offset_channels, kernel_size=3, padding=1)
    conv2 = conv_op(planes, planes, kernel_size=3, padding=1, stride=stride, deformable_groups=deformable_groups,
                    bias=False)
    return conv2_offset, conv2


def _mark_bn_consumers(block):
    """Which convolutions of a residual block are the ONLY consumer of a BatchNorm's output (nn.Conv2d.sole_consumer_of_bn:
    that BatchNorm's backward reductions then ride in the convolution's dgrad epilogue, F.conv2d):
      conv2 (plain)   <- bn1                      conv3 <- bn2 (Bottleneck)
      conv2_offset    <- bn1, forked: the deformable conv takes the alias the offset conv hands back
      conv1           <- the previous block's last BatchNorm, forked: identity shortcut through the alias.  Only blocks
                         without a downsample branch; the block in front is then in the same nn.Sequential and feeds nothing else
                         (a layer's LAST output also leaves the backbone, but it enters the next layer's downsample block)."""
    if hasattr(block, "conv2_offset"):
        block.conv2_offset.sole_consumer_when_forked = True
    elif isinstance(block.conv2, Conv2d):
        block.conv2.sole_consumer_of_bn = True
    if hasattr(block, "conv3"):
        block.conv3.sole_consumer_of_bn = True
    if block.downsample is None:
        block.conv1.sole_consumer_when_forked = True


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dcn=None):
        super(BasicBlock, self).__init__()
        self.with_dcn = dcn is not None
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = bn(planes, fuse_relu=True)
        self.relu = FusedReLU()
        self.with_modulated_dcn = False
        fallback_on_stride = False
        if self.with_dcn:
            fallback_on_stride = dcn.get('fallback_on_stride', False)
            self.with_modulated_dcn = dcn.get('modulated', False)
        if not self.with_dcn or fallback_on_stride:
            self.conv2 = Conv2d(planes, planes, kernel_size=3, padding=1, bias=False)
        else:
            self.conv2_offset, self.conv2 = _make_dcn(dcn, planes, 1)
        self.bn2 = bn(planes, fuse_relu=True)   # ReLU after the residual add
        self.downsample = downsample
        self.stride = stride
        _mark_bn_consumers(self)

    def forward(self, x):
        if self.downsample is None:      # identity shortcut: its gradient is added in conv1's dgrad epilogue (see Bottleneck)
            out, residual = self.conv1.forward_fork(x)
            out = self.bn1(out)
        else:
            residual = x
            out = self.bn1(self.conv1(x))
        if not self.with_dcn:
            out = self.conv2(out)
        elif self.with_modulated_dcn:
            # reference: conv2(out, offset_mask[:, :18], offset_mask[:, -9:].sigmoid()) -- the same expression as one fused
            # autograd node (assets/ops/dcn/deform_conv.py ModulatedDeformConvPackedFunction).  `out` feeds the offset conv AND
            # the deformable conv: the offset conv hands it on as a second output of its own node, so the deformable conv's
            # input gradient is added in the epilogue of the offset conv's dgrad (nn.Conv2d.forward_fork) -- no ATen add
            offset_mask, out = self.conv2_offset.forward_fork(out)
            out = self.conv2.forward_packed(out, offset_mask)
        else:
            out = self.conv2(out, self.conv2_offset(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.bn2(out, residual=residual)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dcn=None):
        super(Bottleneck, self).__init__()
        self.with_dcn = dcn is not None
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = bn(planes, fuse_relu=True)
        fallback_on_stride = False
        self.with_modulated_dcn = False
        if self.with_dcn:
            fallback_on_stride = dcn.get('fallback_on_stride', False)
            self.with_modulated_dcn = dcn.get('modulated', False)
        if not self.with_dcn or fallback_on_stride:
            self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        else:
            self.conv2_offset, self.conv2 = _make_dcn(dcn, planes, stride)
        self.bn2 = bn(planes, fuse_relu=True)
        self.conv3 = Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = bn(planes * 4, fuse_relu=True)   # ReLU after the residual add
        self.relu = FusedReLU()
        self.downsample = downsample
        self.stride = stride
        self.dcn = dcn
        _mark_bn_consumers(self)

    def forward(self, x):
        if self.downsample is None:
            # identity shortcut: x feeds conv1 AND the final add (reference resnet.py:152-181).  conv1 hands x back as a second
            # output of its own node, so the shortcut's gradient is added in the epilogue of conv1's dgrad (nn.Conv2d.forward_fork)
            out, residual = self.conv1.forward_fork(x)
            out = self.bn1(out)
        else:
            residual = x
            out = self.bn1(self.conv1(x))
        if not self.with_dcn:
            out = self.conv2(out)
        elif self.with_modulated_dcn:
            # reference: conv2(out, offset_mask[:, :18], offset_mask[:, -9:].sigmoid()) -- the same expression as one fused
            # autograd node (assets/ops/dcn/deform_conv.py ModulatedDeformConvPackedFunction).  `out` feeds the offset conv AND
            # the deformable conv: the offset conv hands it on as a second output of its own node, so the deformable conv's
            # input gradient is added in the epilogue of the offset conv's dgrad (nn.Conv2d.forward_fork) -- no ATen add
            offset_mask, out = self.conv2_offset.forward_fork(out)
            out = self.conv2.forward_packed(out, offset_mask)
        else:
            out = self.conv2(out, self.conv2_offset(out))
        out = self.bn2(out)
        out = self.conv3(out)
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.bn3(out, residual=residual)   # bn3 + residual add + ReLU in one kernel


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, dcn=None, stage_with_dcn=(False, False, False, False),
                 dilations=[1, 1, 1, 1]):
        self.dcn = dcn
        self.stage_with_dcn = stage_with_dcn
        self.inplanes = 128
        super(ResNet, self).__init__()
        self.conv1 = Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = bn(64, fuse_relu=True)
        self.relu1 = FusedReLU()
        self.conv2 = conv3x3(64, 64)
        self.bn2 = bn(64, fuse_relu=True)
        self.relu2 = FusedReLU()
        self.conv3 = conv3x3(64, 128)
        self.bn3 = bn(128, fuse_relu=True)
        self.relu3 = FusedReLU()
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0], dilation=dilations[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dcn=dcn, dilation=dilations[1])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dcn=dcn, dilation=dilations[2])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dcn=dcn, dilation=dilations[3])
        self.avgpool = nn.AvgPool2d(7, stride=1)                       # unused (reference :210)
        self.fc = Linear(512 * block.expansion, num_classes)           # unused (:211)
        self.smooth = Conv2d(2048, 256, kernel_size=1, stride=1, padding=1)  # unused (:213)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                # draw in logical (OIHW) order like the reference's contiguous weights, then store into the
                # physically-KRSC parameter: identical values for identical seeds
                m.weight.data.copy_(torch.empty(m.weight.shape).normal_(0, math.sqrt(2. / n)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if self.dcn is not None:
            for m in self.modules():
                if isinstance(m, Bottleneck) or isinstance(m, BasicBlock):
                    if hasattr(m, 'conv2_offset'):
                        constant_init(m.conv2_offset, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dcn=None, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False,
                       dilation=dilation),
                bn(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, dcn=dcn)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dcn=dcn))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.bn1(self.conv1(x))
        x = self.bn2(self.conv2(x))
        x = self.bn3(self.conv3(x))
        x = self.maxpool(x)
        x2 = self.layer1(x)
        x3 = self.layer2(x2)
        x4 = self.layer3(x3)
        x5 = self.layer4(x4)
        return x2, x3, x4, x5


def _no_download(pretrained):
    if pretrained:
        raise RuntimeError("pretrained ImageNet weights are downloaded by the reference (resnet.py:13-17); there is "
                           "no network here -- pass pretrained=False / resnet_pretrained=False and load a checkpoint")


def resnet18(pretrained=True, **kwargs):
    _no_download(pretrained)
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34(pretrained=True, **kwargs):
    _no_download(pretrained)
    return ResNet(BasicBlock, [3, 4, 6, 3], **kwargs)


def resnet50(pretrained=True, **kwargs):
    _no_download(pretrained)
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)


def deformable_resnet50(pretrained=True, **kwargs):
    _no_download(pretrained)
    return ResNet(Bottleneck, [3, 4, 6, 3], dcn=dict(modulated=True, deformable_groups=1, fallback_on_stride=False),
                  stage_with_dcn=[False, True, True, True], **kwargs)


def resnet101(pretrained=True, **kwargs):
    _no_download(pretrained)
    return ResNet(Bottleneck, [3, 4, 23, 3], **kwargs)


def resnet152(pretrained=True, **kwargs):
    _no_download(pretrained)
    return ResNet(Bottleneck, [3, 8, 36, 3], **kwargs)
