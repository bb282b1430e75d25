"""Oracle restatement of the reference's ResNet50-dilated + PPM backbone and 2D-CTC decoder on torch CPU ops.

Follows:
  backbones/resnet.py:113-181   Bottleneck (1x1 -> 3x3(stride/dilation) -> 1x1, BN x3, residual add, ReLU)
  backbones/resnet.py:184-256   deep stem (3 x conv3x3, inplanes 128), max-pool 3x3/2, layers [3,4,6,3], init :214-221
  backbones/resnet_dilated.py:5-69  layers 3/4: stride 2 -> 1, 3x3 convs dilated 2 / 4 (first block of each: 1 / 2)
  backbones/ppm.py:6-44         adaptive avg pools {1,2,3,6} -> 1x1 conv+BN+ReLU -> bilinear up -> concat -> conv_last
  decoders/ctc_decoder2d.py:7-53    mask / classify heads, pred = log(max(mask*classify, tiny)) [W,H,N,C], loss / length
  ops/ctc_2d (CUDA only)        -> oracle/ctc2d.py (float64) wrapped as an autograd Function
Module names reproduce the reference's state_dict keys so weights interchange.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ctc2d import ctc2d


class OracleCTC2D(torch.autograd.Function):
    """ops.ctc_loss_2d semantics (ops/ctc_2d/ctc_loss_2d.py:7-37) on CPU via the float64 restatement."""

    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank=0):
        ctx.args = (targets.numpy().copy(), input_lengths.numpy().copy(), target_lengths.numpy().copy(), blank)
        ctx.save_for_backward(log_probs)
        o = ctc2d(log_probs.detach().numpy(), *ctx.args[:3], blank=blank)
        return torch.from_numpy(o['nll']).to(log_probs.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        (log_probs,) = ctx.saved_tensors
        tg, il, tl, blank = ctx.args
        o = ctc2d(log_probs.detach().numpy(), tg, il, tl, blank=blank, grad_out=grad_output.numpy())
        return torch.from_numpy(o['grad']).to(log_probs.dtype), None, None, None, None


oracle_ctc_loss_2d = OracleCTC2D.apply


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None, dcn=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.with_dcn = dcn
        if dcn:  # backbones/resnet.py:125-142: 27-channel offset conv (stride 1 ALWAYS) + modulated deformable conv
            from .dcn import OracleModulatedDeformConv
            self.conv2_offset = nn.Conv2d(planes, 27, 3, padding=1)
            self.conv2 = OracleModulatedDeformConv(planes, planes, 3, padding=1, stride=stride, bias=False)
        else:
            self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        if self.with_dcn:
            om = self.conv2_offset(y)
            y = self.conv2(y, om[:, :18], om[:, -9:].sigmoid())   # resnet.py:162-164
        else:
            y = self.conv2(y)
        y = F.relu(self.bn2(y))
        y = self.bn3(self.conv3(y))
        r = x if self.downsample is None else self.downsample(x)
        return F.relu(y + r)


class _Res50Dilated(nn.Module):
    """resnet50 (deep stem); dilate=True applies the ResnetDilated(dilate_scale=8) rewrite, dilate=False keeps the
    plain stride-32 network together with its unused `fc` / `smooth` parameters (backbones/resnet.py:210-213)."""

    def __init__(self, dilate=True, dcn=False):
        super().__init__()
        self._dcn = dcn
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64)
        self.conv3 = nn.Conv2d(64, 128, 3, 1, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(128)
        self.inplanes = 128
        self.layer1 = self._layer(64, 3, 1)
        self.layer2 = self._layer(128, 4, 2, dcn)
        self.layer3 = self._layer(256, 6, 2, dcn)
        self.layer4 = self._layer(512, 3, 2, dcn)
        # parameters the reference creates (and initialises, consuming RNG) but never uses; ResnetDilated drops them
        fc = nn.Linear(2048, 1000)
        smooth = nn.Conv2d(2048, 256, 1, 1, 1)
        if not dilate:
            self.avgpool = nn.AvgPool2d(7, stride=1)
            self.fc = fc
            self.smooth = smooth
        for m in list(self.modules()) + ([smooth] if dilate else []):
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        del fc
        for m in self.modules():   # resnet.py:222-226: offset convs start at zero
            if isinstance(m, _Bottleneck) and m.with_dcn:
                nn.init.constant_(m.conv2_offset.weight, 0)
                nn.init.constant_(m.conv2_offset.bias, 0)
        if dilate:
            self._dilate(self.layer3, 2)
            self._dilate(self.layer4, 4)

    def _layer(self, planes, blocks, stride, dcn=False):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, down, dcn)]
        self.inplanes = planes * 4
        layers += [_Bottleneck(self.inplanes, planes, dcn=dcn) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    @staticmethod
    def _dilate(layer, dilate):
        for m in layer.modules():
            if isinstance(m, nn.Conv2d):
                if m.stride == (2, 2):
                    m.stride = (1, 1)
                    if m.kernel_size == (3, 3):
                        m.dilation = (dilate // 2, dilate // 2)
                        m.padding = (dilate // 2, dilate // 2)
                elif m.kernel_size == (3, 3):
                    m.dilation = (dilate, dilate)
                    m.padding = (dilate, dilate)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = F.relu(self.bn3(self.conv3(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        x2 = self.layer1(x)
        x3 = self.layer2(x2)
        x4 = self.layer3(x3)
        x5 = self.layer4(x4)
        return x2, x3, x4, x5


class _PPM(nn.Module):
    def __init__(self, inner_channels=256, fc_dim=2048, scales=(1, 2, 3, 6), dropout=0.1):
        super().__init__()
        self.ppm = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(s), nn.Conv2d(fc_dim, 512, 1, bias=False),
                                                nn.BatchNorm2d(512), nn.ReLU()) for s in scales])
        self.cbr_deepsup = nn.Sequential(nn.Conv2d(fc_dim // 2, fc_dim // 4, 3, 1, 1, bias=False),
                                         nn.BatchNorm2d(fc_dim // 4), nn.ReLU())  # unused in forward (ppm.py:20)
        self.conv_last = nn.Sequential(nn.Conv2d(fc_dim + len(scales) * 512, 512, 3, padding=1, bias=False),
                                       nn.BatchNorm2d(512), nn.ReLU(), nn.Dropout2d(dropout),
                                       nn.Conv2d(512, inner_channels, 1))

    def forward(self, conv_out):
        c5 = conv_out[-1]
        size = c5.shape[2:]
        outs = [c5] + [F.interpolate(b(c5), size, mode='bilinear', align_corners=False) for b in self.ppm]
        return self.conv_last(torch.cat(outs, 1))


class Res50PPMBackboneOracle(nn.Sequential):
    """keys `0.*` (ResnetDilated) and `1.*` (PPMDeepsup) like nn.Sequential(resnet_dilated, ppm)."""

    def __init__(self, dropout=0.1):
        super().__init__(_Res50Dilated(), _PPM(dropout=dropout))


class CTCDecoder2DOracle(nn.Module):
    def __init__(self, in_channels=256, num_classes=38, inner_channels=256):
        super().__init__()
        self.pred_mask = nn.Sequential(nn.Identity(), nn.Conv2d(in_channels, inner_channels, 3, padding=1),
                                       nn.Conv2d(inner_channels, 1, 1), nn.Softmax(dim=2))
        self.pred_classify = nn.Sequential(nn.Identity(), nn.Conv2d(in_channels, inner_channels, 3, padding=1),
                                           nn.Conv2d(inner_channels, num_classes, 1))
        self.register_buffer('saved_tiny', torch.tensor(torch.finfo().tiny))

    def forward(self, feature, targets=None, lengths=None, train=False):
        mask = self.pred_mask(feature)
        classify = F.softmax(self.pred_classify(feature), dim=1)
        if self.training:
            pred = torch.log(torch.max(mask * classify, self.saved_tiny)).permute(3, 2, 0, 1).contiguous()
            il = torch.full((feature.shape[0],), pred.shape[0], dtype=torch.long)
            loss = oracle_ctc_loss_2d(pred, targets.long(), il, lengths.long()) / lengths.float()
            return loss, pred
        return classify, mask


class Res50PPM2DCTCOracle(nn.Module):
    def __init__(self, num_classes=38, dropout=0.1):
        super().__init__()
        self.backbone = Res50PPMBackboneOracle(dropout)
        self.decoder = CTCDecoder2DOracle(256, num_classes)

    def forward(self, images, targets=None, lengths=None, train=False):
        return self.decoder(self.backbone(images), targets=targets, lengths=lengths, train=train)


def synthetic_batch_2d(n, height=32, width=64, seed=0, max_label=32, max_len=3, num_classes=38):
    """like oracle.crnn.synthetic_batch but with labels short enough for the W/8 time steps of the 2D-CTC head."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.randint(0, 256, (n, height, width, 3), generator=g, dtype=torch.int64).float()
    mean = torch.tensor([122.67891434, 116.66876762, 104.00698793])
    image = ((pix - mean) / 255.0).permute(0, 3, 1, 2).contiguous()
    length = torch.randint(1, max_len + 1, (n,), generator=g, dtype=torch.int64)
    label = torch.zeros((n, max_label), dtype=torch.int64)
    for i in range(n):
        label[i, :length[i]] = torch.randint(2, num_classes, (int(length[i]),), generator=g)
    return {'image': image, 'label': label.int(), 'length': length.int()}
