"""Oracle restatement of the reference's ResNet50-FPN backbone and attention (Bahdanau GRU) decoder on torch CPU.

Follows:
  backbones/resnet_fpn.py:20-24, fpn_top_down.py:6-30, feature_pyramid.py:4-14   (1x1 laterals, bilinear upsample-add,
      3x3 merge; single stride-4 output map)
  decoders/attention_decoder.py:10-131   encoder (7 conv-bn-relu, pools), one-hot position embeddings, 32-step loop,
      masked NLL with `timestep <= lengths`, teacher forcing fixed by gt_as_output
  decoders/attention_decoder.py:134-231  Attn (Linear(1057->512) on cat([hidden, enc]), tanh, v-dot, softmax),
      AttentionRNNCell (word_linear on one-hot, bmm context, GRUCell, out + log_softmax)
Module names reproduce the reference's state_dict keys.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .res50ppm import _Res50Dilated


class _FPNTopDown(nn.Module):
    def __init__(self, chans=(2048, 1024, 512, 256), feat=256):
        super().__init__()
        self.reduction_layers = nn.ModuleList([nn.Conv2d(c, feat, 1, bias=False) for c in chans])
        self.merge_layer = nn.Conv2d(feat, feat, 3, 1, 1, bias=False)

    def forward(self, feats):
        out = None
        for f, red in zip(feats, self.reduction_layers):
            f = red(f)
            out = f if out is None else F.interpolate(out, size=f.shape[2:], mode='bilinear', align_corners=False) + f
        return self.merge_layer(out)


class Res50FPNOracle(nn.Module):
    def __init__(self):
        super().__init__()
        self.bottom_up = _Res50Dilated(dilate=False)
        self.top_down = _FPNTopDown()

    def forward(self, x):
        return self.top_down(self.bottom_up(x)[::-1])


class _Attn(nn.Module):
    def __init__(self, hidden, embed):
        super().__init__()
        self.attn = nn.Linear(2 * hidden + embed, hidden)
        self.v = nn.Parameter(torch.rand(hidden))
        self.v.data.normal_(mean=0, std=1. / math.sqrt(hidden))

    def forward(self, hidden, enc):            # hidden [N,H], enc [T,N,E]
        T = enc.shape[0]
        e = enc.transpose(0, 1)                # [N,T,E]
        hh = hidden.unsqueeze(1).expand(-1, T, -1)
        energy = torch.tanh(self.attn(torch.cat([hh, e], 2))) @ self.v   # [N,T]
        return F.softmax(energy, dim=1)


class _Cell(nn.Module):
    def __init__(self, hidden, embedded, classes):
        super().__init__()
        self.embedding = nn.Embedding(classes, classes)
        self.embedding.weight.data = torch.eye(classes)
        self.word_linear = nn.Linear(classes, hidden)
        self.attn = _Attn(hidden, embedded)
        self.rnn = nn.GRUCell(2 * hidden + embedded, hidden)
        self.out = nn.Linear(hidden, classes)

    def forward(self, word, hidden, enc, train):
        w_emb = self.word_linear(self.embedding(word.long()))
        a = self.attn(hidden, enc)                                   # [N,T]
        context = torch.bmm(a.unsqueeze(1), enc.transpose(0, 1)).squeeze(1)
        hidden = self.rnn(torch.cat([w_emb, context], 1), hidden)
        o = self.out(hidden)
        return (F.log_softmax(o, 1) if train else F.softmax(o, 1)), hidden, a


class AttentionDecoderOracle(nn.Module):
    def __init__(self, in_channels=256, classes=38, inner=512, max_size=32, height=1, blank=0):
        super().__init__()
        def cbr(i, o, k=3, s=1, p=1):
            return nn.Sequential(nn.Conv2d(i, o, k, s, p), nn.BatchNorm2d(o), nn.ReLU())
        self.encode = nn.Sequential(cbr(in_channels, inner), cbr(inner, inner), nn.MaxPool2d((2, 2), (2, 2)),
                                    cbr(inner, inner), cbr(inner, inner), nn.MaxPool2d((2, 1), (2, 1)),
                                    cbr(inner, inner), cbr(inner, inner), nn.MaxPool2d((2, 1), (2, 1)),
                                    cbr(inner, inner, (2, 3), (2, 1), (0, 1)))
        self.decoder = _Cell(inner, max_size + height, classes)
        self.onehot_embedding_x = nn.Embedding(max_size, max_size)
        self.onehot_embedding_x.weight.data = torch.eye(max_size)
        self.onehot_embedding_y = nn.Embedding(height, height)
        self.onehot_embedding_y.weight.data = torch.eye(height)
        self.inner, self.max_size, self.height, self.blank = inner, max_size, height, blank

    def forward(self, feature, targets=None, lengths=None, train=False, coins=None):
        """coins (test infrastructure, optional list of max_size bools): the teacher-forcing decisions of the training loop,
        reference attention_decoder.py:107-110 `if self._get_gt_as_output(): timestep_input = targets[:, timestep] else:
        timestep_input = i.detach()`; None = gt_as_output=True (what the pinned goldens use)."""
        seq = self.encode(feature)                                    # [N,512,1,32]
        N = feature.shape[0]
        dev = feature.device      # (device-agnostic so that tools/bench_reference_stack_gpu.py can time these modules on the GPU)
        iy, ix = torch.meshgrid(torch.arange(self.height, device=dev), torch.arange(self.max_size, device=dev), indexing='ij')
        ex = self.onehot_embedding_x(ix).permute(2, 0, 1).unsqueeze(0).expand(N, -1, -1, -1)
        ey = self.onehot_embedding_y(iy).permute(2, 0, 1).unsqueeze(0).expand(N, -1, -1, -1)
        dec_in = torch.cat([seq, ey, ex], 1).reshape(N, -1, self.height * self.max_size).permute(2, 0, 1)  # [T,N,545]
        hidden = torch.zeros(N, self.inner, dtype=feature.dtype, device=dev)
        word = torch.full((N,), self.blank, dtype=torch.long, device=dev)
        if self.training:
            targets = targets.long()
            loss, atts = 0, []
            for t in range(self.max_size):
                out, hidden, a = self.decoder(word, hidden, dec_in, True)
                loss = loss + F.nll_loss(out, targets[:, t], reduction='none') * (t <= lengths).float()
                atts.append(a.unsqueeze(1))
                word = targets[:, t] if (coins is None or coins[t]) else out.argmax(1).detach()   # :107-110
            return loss, torch.cat(atts, 1).view(N, -1, self.height, self.max_size)
        pred = torch.full((N, self.max_size), self.blank, dtype=torch.int32, device=dev)
        for t in range(self.max_size):
            out, hidden, a = self.decoder(word, hidden, dec_in, False)
            word = out.argmax(1)
            pred[:, t] = word
            if (word == self.blank).all():
                break
        return pred


class FPNAttentionOracle(nn.Module):
    def __init__(self, classes=38):
        super().__init__()
        self.backbone = Res50FPNOracle()
        self.decoder = AttentionDecoderOracle(256, classes)

    def forward(self, images, targets=None, lengths=None, train=False):
        return self.decoder(self.backbone(images), targets=targets, lengths=lengths, train=train)
