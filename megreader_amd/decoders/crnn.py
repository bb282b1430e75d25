"""CRNN sequence decoder (2 x BiLSTM + Linear, fused log-softmax + CTC) on HIP kernels.

Mirror of reference decoders/crnn.py:8-104: same constructor signature, parameter names
(``rnn.{0,1}.rnn.weight_ih_l0`` ..., ``rnn.{0,1}.embedding.{weight,bias}``) and return values
(training: ``(loss f64 scalar, pred f64 [T,N,C] log-probabilities)``; eval: ``[N,C,1,T]`` softmax).
"""
import torch
import torch.nn as nn

from ..charsets import DefaultCharset
from ..nn import LSTM, Linear
from ..nn import functional as F


class BidirectionalLSTM(nn.Module):

    def __init__(self, nIn, nHidden, nOut):
        super(BidirectionalLSTM, self).__init__()
        self.rnn = LSTM(nIn, nHidden, bidirectional=True)
        self.embedding = Linear(nHidden * 2, nOut)

    def forward(self, input):
        recurrent, _ = self.rnn(input)      # [T, b, 2H]
        return self.embedding(recurrent)     # [T, b, nOut]


class CRNNDecoder(nn.Module):

    def __init__(self, charset=DefaultCharset(), inner_channels=256, in_channels=256, need_reduce=False,
                 reduce_func=None, loss_func='pytorch'):
        super().__init__()
        if need_reduce:
            raise NotImplementedError("CRNNDecoder(need_reduce=True) is not used by any reference experiment")
        if loss_func != 'pytorch':
            raise NotImplementedError("only loss_func='pytorch' (nn.CTCLoss semantics) is implemented")
        self.rnn = nn.Sequential(
            BidirectionalLSTM(in_channels, inner_channels, inner_channels),
            BidirectionalLSTM(inner_channels, inner_channels, len(charset)))
        self.inner_channels = inner_channels
        self.blank = getattr(charset, 'blank', 0)

    def forward(self, feature, targets=None, lengths=None, train=False):
        b, c, h, w = feature.size()
        assert h == 1, "the height of conv must be 1"
        seq = F.map_to_sequence(feature)     # [W, N, C]
        pred = self.rnn(seq)                 # [W, N, classes] logits

        if train:
            # reference: log_softmax(dim=2).to(float64) -> nn.CTCLoss(zero_infinity=True) with input_lengths = T
            loss, log_probs = F.ctc_loss_logits(pred, targets, None, lengths, blank=0, zero_infinity=True)
            return loss, log_probs.to(torch.float64)
        else:
            return F.softmax_eval_nc1t(pred)
