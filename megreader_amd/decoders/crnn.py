"""CRNN sequence decoder (2 x BiLSTM + Linear, fused log-softmax + CTC) on HIP kernels.

Mirror of reference decoders/crnn.py:8-104: same constructor signature, parameter names
(``rnn.{0,1}.rnn.weight_ih_l0`` ..., ``rnn.{0,1}.embedding.{weight,bias}``) and return values
(training: ``(loss f64 scalar, pred f64 [T,N,C] log-probabilities)``; eval: ``[N,C,1,T]`` softmax).
"""
import torch
import torch.nn as nn

from ..charsets import DefaultCharset
from ..nn import LSTM, BatchNorm2d, Conv2d, FusedReLU, Linear, MaxPool2d
from ..nn import functional as F


class BidirectionalLSTM(nn.Module):

    def __init__(self, nIn, nHidden, nOut):
        super(BidirectionalLSTM, self).__init__()
        self.rnn = LSTM(nIn, nHidden, bidirectional=True)
        self.embedding = Linear(nHidden * 2, nOut)

    def forward(self, input):
        recurrent, _ = self.rnn(input)      # [T, b, 2H]
        return self.embedding(recurrent)     # [T, b, nOut]


class _MaxOverHeight(nn.Module):
    """nn.AdaptiveMaxPool2d((1, None)) (reference decoders/crnn.py:67-68): max over the whole height, width kept."""

    def forward(self, x):
        h = x.size(2)
        return F.max_pool2d(x, (h, 1), (h, 1), (0, 0))


class CRNNDecoder(nn.Module):

    def __init__(self, charset=DefaultCharset(), inner_channels=256, in_channels=256, need_reduce=False,
                 reduce_func=None, loss_func='pytorch'):
        super().__init__()
        rnn_input = inner_channels if need_reduce else in_channels
        self.rnn = nn.Sequential(
            BidirectionalLSTM(rnn_input, inner_channels, inner_channels),
            BidirectionalLSTM(inner_channels, inner_channels, len(charset)))
        self.inner_channels = inner_channels
        if need_reduce:     # reference decoders/crnn.py:42-46: map an [N,C,h>1,W] feature to height 1
            if reduce_func == 'conv':
                self.fpn2rnn = self._init_conv(in_channels)
            elif reduce_func == 'pooling':
                self.fpn2rnn = self._init_pooling()
        # 'pytorch': nn.CTCLoss(zero_infinity=True) -> f64 scalar; anything else: the reference's own python CTCLoss
        # (decoders/ctc_loss.py) -> per-sample nll / target_length, [N]
        self.per_sample_loss = loss_func != 'pytorch'
        self.blank = getattr(charset, 'blank', 0)

    def _init_conv(self, in_channels, stride=(2, 1), padding=(0, 1)):
        return nn.Sequential(
            self.conv_bn_relu(in_channels, self.inner_channels),
            MaxPool2d((2, 2), (2, 2), (0, 0)),
            self.conv_bn_relu(self.inner_channels, self.inner_channels),
            MaxPool2d(stride, stride, (0, 0)),
            self.conv_bn_relu(self.inner_channels, self.inner_channels),
            MaxPool2d(stride, stride, (0, 0)))

    def _init_pooling(self):
        return _MaxOverHeight()

    def conv_bn_relu(self, input_channels, output_channels, kernel_size=3, stride=1, padding=1):
        return nn.Sequential(
            Conv2d(input_channels, output_channels, kernel_size=kernel_size, stride=stride, padding=padding),
            BatchNorm2d(output_channels, fuse_relu=True), FusedReLU())

    def forward(self, feature, targets=None, lengths=None, train=False):
        b, c, h, w = feature.size()
        if h > 1:
            feature = self.fpn2rnn(feature)
            b, c, h, w = feature.size()
        assert h == 1, "the height of conv must be 1"
        seq = F.map_to_sequence(feature)     # [W, N, C]
        pred = self.rnn(seq)                 # [W, N, classes] logits

        if train:
            # reference: log_softmax(dim=2).to(float64) -> self.ctc_loss(pred, targets, [T]*N, lengths)
            loss, log_probs = F.ctc_loss_logits(pred, targets, None, lengths, blank=0,
                                                zero_infinity=not self.per_sample_loss,
                                                per_sample=self.per_sample_loss, log_probs_f64=True)
            return loss, log_probs
        else:
            return F.softmax_eval_nc1t(pred)
