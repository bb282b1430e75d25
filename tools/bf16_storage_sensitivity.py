#!/usr/bin/env python
"""How far do the gradients of the reference's models move when every activation (and every activation gradient) is merely
STORED in bfloat16 -- float64 arithmetic everywhere, only the tensors between layers rounded to 8 mantissa bits?

CPU only, the oracle restatements of the reference models (oracle/), nothing of the product.  Purpose: the timed-step tests
(tests/test_timed_step_gpu.py) report the bf16 gradients of the replayed HIP step against the float64 oracle per parameter.  For the
53-BatchNorm ResNet-50 models at random initialisation on synthetic crops that comparison comes out near-orthogonal (relative L2
~1.35, cosine ~0.08 for EVERY backbone parameter, profiles/r05_bf16_drift_timed_step.txt) while the same kernels in float32 agree
with float64 to the bars of tests/_parity.py.  This script answers whether that is the kernels or the problem: it perturbs the
float64 oracle itself with bf16 storage and prints the same two numbers.

usage: python tools/bf16_storage_sensitivity.py [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _Round(torch.autograd.Function):
    """y = bf16(x) forward, dx = bf16(dy) backward (straight-through: the rounding itself has no gradient)."""

    @staticmethod
    def forward(ctx, x):
        return bf16_round(x)

    @staticmethod
    def backward(ctx, g):
        return bf16_round(g)


def store_in_bf16(model):
    """Round the output of every leaf module that produces an activation tensor (conv, BatchNorm, Linear, ReLU, pooling, LSTM)."""
    handles = []

    def hook(_m, _inp, out):
        if isinstance(out, torch.Tensor) and out.is_floating_point() and out.requires_grad:
            return _Round.apply(out)
        if isinstance(out, tuple) and out and isinstance(out[0], torch.Tensor) and out[0].requires_grad:
            return (_Round.apply(out[0]),) + tuple(out[1:])
        return None
    for m in model.modules():
        if not list(m.children()):
            handles.append(m.register_forward_hook(hook))
    return handles


def grads_of(model, loss_fn):
    for p in model.parameters():
        p.grad = None
    loss = loss_fn()
    loss.backward()
    return float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def compare(name, model, loss_fn, groups):
    l0, g0 = grads_of(model, loss_fn)
    handles = store_in_bf16(model)
    l1, g1 = grads_of(model, loss_fn)
    for h in handles:
        h.remove()
    print("%s: loss float64 %.6f, with bf16 storage %.6f" % (name, l0, l1))
    for label, pred in groups:
        l2s, coss = [], []
        for k in g0:
            if not pred(k) or float(g0[k].abs().max()) < 1e-9:
                continue
            a, b = g0[k].flatten(), g1[k].flatten()
            l2s.append(float((a - b).norm() / a.norm()))
            coss.append(float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)))
        if l2s:
            print("   %-28s %3d parameters: relative L2 median %.3f (max %.3f), cosine median %.4f (min %.4f)" %
                  (label, len(l2s), sorted(l2s)[len(l2s) // 2], max(l2s), sorted(coss)[len(coss) // 2], min(coss)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    torch.manual_seed(99)
    from oracle.crnn import CRNNOracle, synthetic_batch
    from oracle.res50ppm import Res50PPM2DCTCOracle, synthetic_batch_2d
    crnn = CRNNOracle().double().train()
    b = synthetic_batch(a.batch, 32, 128, seed=5)
    img = b['image'].double()
    compare("CRNN, %d crops of 32x128" % a.batch, crnn,
            lambda: crnn(img, targets=b['label'], lengths=b['length'].long(), train=True)[0].mean(),
            [("conv backbone", lambda k: k.startswith("backbone")), ("BiLSTM head", lambda k: k.startswith("decoder"))])
    torch.manual_seed(99)
    res = Res50PPM2DCTCOracle(dropout=0.0).double().train()
    b2 = synthetic_batch_2d(a.batch, 32, 128, seed=5, max_len=3)
    img2 = b2['image'].double()
    compare("ResNet50-dilated-PPM + 2D-CTC, %d crops of 32x128" % a.batch, res,
            lambda: res(img2, targets=b2['label'], lengths=b2['length'].long(), train=True)[0].mean(),
            [("stem + layer1", lambda k: k.startswith("backbone.0.conv") or k.startswith("backbone.0.bn") or ".layer1." in k),
             ("layer2", lambda k: ".layer2." in k), ("layer3", lambda k: ".layer3." in k), ("layer4", lambda k: ".layer4." in k),
             ("PPM + heads", lambda k: k.startswith("backbone.1") or k.startswith("decoder"))])


if __name__ == "__main__":
    main()
