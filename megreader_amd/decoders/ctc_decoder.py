"""Conv-encoder + 1-D CTC head on HIP kernels -- mirror of reference decoders/ctc_decoder.py:13-66 (`CTCDecoder`, exported at
decoders/__init__.py): same constructor, parameter names (`encode.{0,1,3,4,6,7,9}.{0,1}.*`, `pred_conv.{weight,bias}`) and
default initialisation order, same forward contract (training: `(loss f32 scalar, log-probs [N, C, W])`, eval: softmax
`[N, C, 1, W]`).

The seven conv-BN-ReLU stages and the three max-pools are the HIP layers (the same encoder as the attention decoder's,
attention_decoder.py:36-49); `pred_conv` (1x1, bias) runs as an MFMA GEMM over the [W*N, inner] sequence; log-softmax + CTC
(alpha / beta / gradient) is the fused kernel of the CRNN head (csrc/ctc.hip) with the reference's settings here:
nn.CTCLoss(reduction='mean') WITHOUT zero_infinity, input lengths fixed at 32 (ctc_decoder.py:17, 61-62)."""
import torch
import torch.nn as nn

from ..charsets import DefaultCharset
from ..nn import BatchNorm2d, Conv2d, FusedReLU, MaxPool2d
from ..nn import functional as F


class CTCDecoder(nn.Module):
    def __init__(self, in_channels, charset=DefaultCharset(), inner_channels=256, **kwargs):
        super(CTCDecoder, self).__init__()
        self.inner_channels = inner_channels
        self.encode = self._init_encoder(in_channels)
        # parameter holder with nn.Conv2d's names / shapes / init (weight [classes, inner, 1, 1]); applied as a GEMM
        self.pred_conv = nn.Conv2d(inner_channels, len(charset), kernel_size=1, bias=True, padding=0)
        self.blank = 0
        if 'blank' in kwargs:
            self.blank = kwargs['blank']

    def _init_encoder(self, in_channels, stride=(2, 1), padding=(0, 1)):
        c = self.inner_channels
        return nn.Sequential(
            self.conv_bn_relu(in_channels, c), self.conv_bn_relu(c, c), MaxPool2d((2, 2), (2, 2), (0, 0)),
            self.conv_bn_relu(c, c), self.conv_bn_relu(c, c), MaxPool2d(stride, stride, (0, 0)),
            self.conv_bn_relu(c, c), self.conv_bn_relu(c, c), MaxPool2d(stride, stride, (0, 0)),
            self.conv_bn_relu(c, c, kernel_size=(2, 3), stride=stride, padding=padding))

    def conv_bn_relu(self, input_channels, output_channels, kernel_size=3, stride=1, padding=1):
        return nn.Sequential(Conv2d(input_channels, output_channels, kernel_size=kernel_size, stride=stride,
                                    padding=padding),
                             BatchNorm2d(output_channels, fuse_relu=True), FusedReLU())

    def forward(self, feature, targets=None, lengths=None, train=False):
        if not feature.is_cuda:
            raise NotImplementedError("megreader_amd decoders run on the GPU only")
        enc = self.encode(feature)                       # logical [N, inner, h, W]
        N, C, h, W = enc.shape
        w = self.pred_conv.weight

        def row_logits(r):
            seq = F.map_to_sequence(enc if h == 1 else enc[:, :, r:r + 1, :])               # [W, N, inner]
            return F.linear(seq, w.reshape(w.shape[0], w.shape[1]), self.pred_conv.bias)     # [W, N, classes]

        if not train and h != 1:
            # eval: the reference returns the softmax of EVERY row, [N, classes, h, W] (ctc_decoder.py:64-66)
            return torch.cat([F.softmax_eval_nc1t(row_logits(r)) for r in range(h)], dim=2)
        # training: the reference applies the 1x1 conv to every row and then `select(2, 0)` (ctc_decoder.py:57-60): row 0 counts
        logits = row_logits(0)
        if train:
            if W < 32:
                raise RuntimeError("CTCDecoder: the reference fixes input_lengths at 32 (ctc_decoder.py:61) but the encoder "
                                   "output has only %d columns" % W)
            il = None if W == 32 else torch.full((N,), 32, dtype=torch.int64, device=feature.device)
            # nn.CTCLoss(reduction='mean') with its default blank = 0 (ctc_decoder.py:17): `self.blank` is stored by the
            # reference but never reaches the loss -- it only matters to the representer
            loss, log_probs = F.ctc_loss_logits(logits, targets, il, lengths, blank=0, zero_infinity=False)
            return loss.to(torch.float32), log_probs.to(torch.float32).permute(1, 2, 0)
        return F.softmax_eval_nc1t(logits)
