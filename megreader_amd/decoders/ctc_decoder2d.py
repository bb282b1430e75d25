"""2D-CTC recognition head on HIP kernels.

Mirror of reference decoders/ctc_decoder2d.py:7-53: same constructor, parameter names (`pred_mask.{1,2}`,
`pred_classify.{1,2}`, buffer `saved_tiny`), same branching on `self.training` (not on the `train` argument) and the
same return values: training `(loss[N] = ctc_loss_2d(pred, ...) / lengths, pred[W,H,N,C])`, eval `(classify, mask)`.
softmax over H, softmax over C, product, clamp at `tiny`, log and the [W,H,N,C] permute run in one kernel.
"""
import torch
import torch.nn as nn

from ..charsets import DefaultCharset
from ..nn import Conv2d
from ..nn import functional as F


class _Identity(nn.Module):
    def forward(self, x):
        return x


class CTCDecoder2D(nn.Module):
    def __init__(self, in_channels, charset=DefaultCharset(), inner_channels=256, stride=1, blank=0, **kwargs):
        super(CTCDecoder2D, self).__init__()
        if stride != 1:
            raise NotImplementedError("CTCDecoder2D(stride != 1) is not used by any reference experiment")
        self.charset = charset
        from ..ops import ctc_loss_2d
        self.ctc_loss = ctc_loss_2d
        self.inner_channels = inner_channels
        # index 0 is nn.AvgPool2d(kernel_size=1, stride=1) in the reference == identity; index 3 the Softmax(dim=2)
        self.pred_mask = nn.Sequential(
            _Identity(),
            Conv2d(in_channels, inner_channels, kernel_size=3, padding=1),
            Conv2d(inner_channels, 1, kernel_size=1),
            _Identity())
        self.pred_classify = nn.Sequential(
            _Identity(),
            Conv2d(in_channels, inner_channels, kernel_size=3, padding=1),
            Conv2d(inner_channels, len(charset), kernel_size=1))
        self.blank = blank
        self.tiny = torch.tensor(torch.finfo().tiny, requires_grad=False)
        self.register_buffer('saved_tiny', self.tiny)

    def forward(self, feature, targets=None, lengths=None, train=False, masks=None, segs=None):
        if isinstance(feature, tuple):
            feature = feature[-1]
        mask_logits = self.pred_mask(feature)          # [N,1,H,W] logits (softmax over H is fused below)
        cls_logits = self.pred_classify(feature)       # [N,C,H,W] logits
        pred, mask, classify = F.ctc2d_head(mask_logits, cls_logits, float(torch.finfo().tiny))
        if self.training:
            n = feature.size()[0]
            input_lengths = torch.full((n,), pred.shape[0], dtype=torch.long, device=pred.device)
            loss = self.ctc_loss(pred, targets.long(), input_lengths, lengths.long()) / lengths.float()
            return loss, pred
        else:
            return classify, mask
