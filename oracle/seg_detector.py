"""Oracle restatement of the DB head and its loss on torch CPU ops -- reference decoders/seg_detector.py:7-147
(SegDetector with adaptive thresholding, smooth = serial = False) and decoders/seg_detector_loss.py:157-185
(L1BalanceCELoss = balance_cross_entropy_loss.py:29-56 + l1_loss.py:5-11 + dice_loss.py:28-42).  Pinned to the unmodified
reference modules (same seeded initialisation, bit-identical outputs) by tests/test_oracle_models.py."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class SegDetectorOracle(nn.Module):
    def __init__(self, in_channels=(64, 128, 256, 512), inner_channels=256, k=10, bias=False, adaptive=False):
        super().__init__()
        self.k, self.adaptive = k, adaptive
        q = inner_channels // 4
        self.up5 = nn.Upsample(scale_factor=2, mode='nearest')
        self.up4 = nn.Upsample(scale_factor=2, mode='nearest')
        self.up3 = nn.Upsample(scale_factor=2, mode='nearest')
        self.in5 = nn.Conv2d(in_channels[-1], inner_channels, 1, bias=bias)
        self.in4 = nn.Conv2d(in_channels[-2], inner_channels, 1, bias=bias)
        self.in3 = nn.Conv2d(in_channels[-3], inner_channels, 1, bias=bias)
        self.in2 = nn.Conv2d(in_channels[-4], inner_channels, 1, bias=bias)
        self.out5 = nn.Sequential(nn.Conv2d(inner_channels, q, 3, padding=1, bias=bias),
                                  nn.Upsample(scale_factor=8, mode='nearest'))
        self.out4 = nn.Sequential(nn.Conv2d(inner_channels, q, 3, padding=1, bias=bias),
                                  nn.Upsample(scale_factor=4, mode='nearest'))
        self.out3 = nn.Sequential(nn.Conv2d(inner_channels, q, 3, padding=1, bias=bias),
                                  nn.Upsample(scale_factor=2, mode='nearest'))
        self.out2 = nn.Conv2d(inner_channels, q, 3, padding=1, bias=bias)
        self.binarize = self._head(inner_channels, bias)
        self.binarize.apply(self.weights_init)
        if adaptive:
            self.thresh = self._head(inner_channels, bias)
            self.thresh.apply(self.weights_init)
        for m in (self.in5, self.in4, self.in3, self.in2, self.out5, self.out4, self.out3, self.out2):
            m.apply(self.weights_init)

    @staticmethod
    def _head(inner_channels, bias):
        q = inner_channels // 4
        return nn.Sequential(nn.Conv2d(inner_channels, q, 3, padding=1, bias=bias), nn.BatchNorm2d(q),
                             nn.ReLU(inplace=True), nn.ConvTranspose2d(q, q, 2, 2), nn.BatchNorm2d(q),
                             nn.ReLU(inplace=True), nn.ConvTranspose2d(q, 1, 2, 2), nn.Sigmoid())

    @staticmethod
    def weights_init(m):
        name = m.__class__.__name__
        if name.find('Conv') != -1:
            nn.init.kaiming_normal_(m.weight.data)
        elif name.find('BatchNorm') != -1:
            m.weight.data.fill_(1.)
            m.bias.data.fill_(1e-4)

    def forward(self, features):
        c2, c3, c4, c5 = features
        in5, in4, in3, in2 = self.in5(c5), self.in4(c4), self.in3(c3), self.in2(c2)
        out4 = self.up5(in5) + in4
        out3 = self.up4(out4) + in3
        out2 = self.up3(out3) + in2
        fuse = torch.cat((self.out5(in5), self.out4(out4), self.out3(out3), self.out2(out2)), 1)
        binary = self.binarize(fuse)
        result = OrderedDict(binary=binary)
        if self.adaptive:
            thresh = self.thresh(fuse)
            result.update(thresh=thresh, thresh_binary=torch.reciprocal(1 + torch.exp(-self.k * (binary - thresh))))
        return result


def l1_balance_ce_loss(pred, batch, eps=1e-6, l1_scale=10, bce_scale=5, negative_ratio=3.0):
    gt, mask = batch['gt'], batch['mask']
    positive = (gt * mask).byte()
    negative = ((1 - gt) * mask).byte()
    pc = int(positive.float().sum())
    nc = min(int(negative.float().sum()), int(pc * negative_ratio))
    bce = F.binary_cross_entropy(pred['binary'], gt, reduction='none')[:, 0]
    neg, _ = torch.topk((bce * negative.float()).view(-1), nc)
    bce_loss = ((bce * positive.float()).sum() + neg.sum()) / (pc + nc + eps)
    l1 = (torch.abs(pred['thresh'][:, 0] - batch['thresh_map']) * batch['thresh_mask']).sum() / batch['thresh_mask'].sum()
    p, g = pred['thresh_binary'][:, 0], gt[:, 0]
    dice = 1 - 2.0 * (p * g * mask).sum() / ((p * mask).sum() + (g * mask).sum() + eps)
    return dice + l1_scale * l1 + bce_loss * bce_scale
