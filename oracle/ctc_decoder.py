"""Oracle restatement of the reference's conv-encoder + 1-D CTC head on torch CPU ops -- decoders/ctc_decoder.py:13-66
(`CTCDecoder`: seven conv-BN-ReLU stages with three max-pools, a 1x1 prediction conv, LogSoftmax(dim=1), select row 0,
nn.CTCLoss(reduction='mean') with input lengths fixed at 32).  Module names reproduce the reference's state_dict keys.
PINNED: bit-identical to the unmodified reference module on CPU (tests/test_oracle_models.py::test_ctc_decoder_oracle...)."""
import torch
import torch.nn as nn


class CTCDecoderOracle(nn.Module):
    def __init__(self, in_channels, num_classes=38, inner_channels=256, blank=0):
        super().__init__()
        self.ctc_loss = nn.CTCLoss(reduction='mean')

        def cbr(i, o, k=3, s=1, p=1):
            return nn.Sequential(nn.Conv2d(i, o, kernel_size=k, stride=s, padding=p), nn.BatchNorm2d(o), nn.ReLU(inplace=True))
        c = inner_channels
        self.encode = nn.Sequential(cbr(in_channels, c), cbr(c, c), nn.MaxPool2d((2, 2), (2, 2), (0, 0)),
                                    cbr(c, c), cbr(c, c), nn.MaxPool2d((2, 1), (2, 1), (0, 0)),
                                    cbr(c, c), cbr(c, c), nn.MaxPool2d((2, 1), (2, 1), (0, 0)),
                                    cbr(c, c, (2, 3), (2, 1), (0, 1)))
        self.pred_conv = nn.Conv2d(c, num_classes, kernel_size=1, bias=True, padding=0)
        self.blank = blank

    def forward(self, feature, targets=None, lengths=None, train=False):
        pred = self.pred_conv(self.encode(feature))
        if train:
            pred = torch.log_softmax(pred, dim=1).select(2, 0).permute(2, 0, 1)     # W, N, C
            input_lengths = torch.zeros((feature.size()[0],), dtype=torch.int) + 32
            loss = self.ctc_loss(pred, targets, input_lengths, lengths)
            return loss, pred.permute(1, 2, 0)
        return torch.softmax(pred, dim=1)
