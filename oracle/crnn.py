"""Oracle restatement of the reference CRNN recognition model on torch CPU ops.

Follows:
  backbones/crnn.py:4-63     7 conv stages, BN stages have NO activation, pools (2,2),(2,2),(2,2)/(2,1)/(0,1) x2
  decoders/crnn.py:8-24      BidirectionalLSTM = nn.LSTM(bidirectional) + nn.Linear
  decoders/crnn.py:80-104    squeeze/permute, 2 x BiLSTM, train: log_softmax -> float64 -> nn.CTCLoss(zero_infinity)
                             with input_lengths = T for every sample; eval: [N,C,1,T] softmax over C
  structure/model.py:16-24   BasicModel = decoder(backbone(x), ...)
  trainer.py:114-130         train step = zero_grad, forward, loss.mean(), backward, optimizer.step()
The module tree reproduces the reference's state_dict keys so weights interchange.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# (out_channels, kernel, pad, batch_norm, pool) per stage; pool = None | (kernel, stride, padding)
_STAGES = [
    (64, 3, 1, False, ((2, 2), (2, 2), (0, 0))),
    (128, 3, 1, False, ((2, 2), (2, 2), (0, 0))),
    (256, 3, 1, True, None),
    (256, 3, 1, False, ((2, 2), (2, 1), (0, 1))),
    (512, 3, 1, True, None),
    (512, 3, 1, False, ((2, 2), (2, 1), (0, 1))),
    (512, 2, 0, True, None),
]


class CRNNBackboneOracle(nn.Module):
    def __init__(self, nc=3):
        super().__init__()
        stages = []
        cin = nc
        for cout, k, p, bn, pool in _STAGES:
            body = nn.Sequential(nn.Conv2d(cin, cout, k, 1, p), nn.BatchNorm2d(cout) if bn else nn.ReLU())
            stages.append(body if pool is None else nn.Sequential(body, nn.MaxPool2d(*pool)))
            cin = cout
        self.cnn = nn.Sequential(*stages)

    def forward(self, x):
        return self.cnn(x)


class _BiLSTMHead(nn.Module):
    def __init__(self, n_in, n_hidden, n_out):
        super().__init__()
        self.rnn = nn.LSTM(n_in, n_hidden, bidirectional=True)
        self.embedding = nn.Linear(2 * n_hidden, n_out)

    def forward(self, x):
        h, _ = self.rnn(x)
        t, b, c = h.shape
        return self.embedding(h.reshape(t * b, c)).view(t, b, -1)


class CRNNDecoderOracle(nn.Module):
    """decoders/crnn.py:27-104.  need_reduce / reduce_func (:42-46, :52-68): 'conv' = three conv-BN-ReLU + max-pool stages
    that halve the height ((2,2) then (2,1) twice), 'pooling' = max over the height.  loss_func != 'pytorch' selects the
    reference's own python CTC (decoders/ctc_loss.py), restated here as what it computes: per-sample nll / target length
    (its 'mean' reduction, :118-122), no zero_infinity."""

    def __init__(self, num_classes=38, inner_channels=256, in_channels=512, need_reduce=False, reduce_func=None,
                 loss_func='pytorch'):
        super().__init__()
        rnn_input = inner_channels if need_reduce else in_channels
        self.rnn = nn.Sequential(_BiLSTMHead(rnn_input, inner_channels, inner_channels),
                                 _BiLSTMHead(inner_channels, inner_channels, num_classes))
        if need_reduce and reduce_func == 'conv':
            def cbr(i, o):
                return nn.Sequential(nn.Conv2d(i, o, 3, 1, 1), nn.BatchNorm2d(o), nn.ReLU(inplace=True))
            self.fpn2rnn = nn.Sequential(cbr(in_channels, inner_channels), nn.MaxPool2d((2, 2), (2, 2), (0, 0)),
                                         cbr(inner_channels, inner_channels), nn.MaxPool2d((2, 1), (2, 1), (0, 0)),
                                         cbr(inner_channels, inner_channels), nn.MaxPool2d((2, 1), (2, 1), (0, 0)))
        elif need_reduce and reduce_func == 'pooling':
            self.fpn2rnn = nn.AdaptiveMaxPool2d((1, None))
        self.per_sample_loss = loss_func != 'pytorch'

    def logits(self, feature):
        if feature.shape[2] > 1:
            feature = self.fpn2rnn(feature)
        assert feature.shape[2] == 1
        return self.rnn(feature.squeeze(2).permute(2, 0, 1))  # [W, N, classes]

    def forward(self, feature, targets=None, lengths=None, train=False):
        pred = self.logits(feature)
        if train:
            logp = F.log_softmax(pred, dim=2).to(torch.float64)
            t, b = logp.shape[0], logp.shape[1]
            in_len = torch.full((b,), t, dtype=torch.int32)
            if self.per_sample_loss:
                nll = F.ctc_loss(logp, targets, in_len, lengths, blank=0, reduction='none', zero_infinity=False)
                return nll / lengths.to(torch.float64), logp
            loss = F.ctc_loss(logp, targets, in_len, lengths, blank=0, reduction='mean', zero_infinity=True)
            return loss, logp
        return F.softmax(pred.permute(1, 2, 0).unsqueeze(2), dim=1)


class CRNNOracle(nn.Module):
    """BasicModel(crnn_backbone, CRNNDecoder) of the reference: keys `backbone.*`, `decoder.*`."""

    def __init__(self, num_classes=38, nc=3):
        super().__init__()
        self.backbone = CRNNBackboneOracle(nc)
        self.decoder = CRNNDecoderOracle(num_classes)

    def forward(self, images, targets=None, lengths=None, train=False):
        return self.decoder(self.backbone(images), targets=targets, lengths=lengths, train=train)


def synthetic_batch(n, height=32, width=128, seed=0, max_label=32, min_len=3, max_len=10, num_classes=38):
    """Synthetic batch per BASELINE.md §3: uint8 pixels -> (x - mean)/255 (data/processes/normalize_image.py:8-16),
    labels of length U{min_len..max_len} over class ids U{2..C-1}, zero padded to 32 (int32), length int32."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.randint(0, 256, (n, height, width, 3), generator=g, dtype=torch.int64).float()
    mean = torch.tensor([122.67891434, 116.66876762, 104.00698793])
    image = ((pix - mean) / 255.0).permute(0, 3, 1, 2).contiguous()
    length = torch.randint(min_len, max_len + 1, (n,), generator=g, dtype=torch.int64)
    label = torch.zeros((n, max_label), dtype=torch.int64)
    for i in range(n):
        label[i, :length[i]] = torch.randint(2, num_classes, (int(length[i]),), generator=g)
    return {'image': image, 'label': label.int(), 'length': length.int()}


def train_step(model, optimizer, batch):
    """reference trainer.py:114-130 on CPU.  Returns the loss value."""
    optimizer.zero_grad()
    loss, _ = model(batch['image'], targets=batch['label'], lengths=batch['length'].long(), train=True)
    loss = loss.mean()
    loss.backward()
    optimizer.step()
    return float(loss.detach())
