#!/usr/bin/env python
"""Localise differences between the persistent and the per-step LSTM recurrence: same inputs through both C entry
points, per (step, direction, unit slice) max error of out / cbuf / gates (forward) and dgates (backward)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megreader_amd._lib import call, load, ptr  # noqa: E402


def main():
    T, N, H = int(os.environ.get("T", 33)), int(os.environ.get("N", 256)), 256
    lib = load()
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    bf = torch.bfloat16
    xproj = (torch.randn(T * N, 8 * H, generator=g) * 0.5).to(bf).to(dev)
    whh = (torch.rand(2, 4 * H, H, generator=g) * 0.125 - 0.0625).to(bf).to(dev)
    whh_t = whh.transpose(1, 2).contiguous()
    dout = torch.randn(T, N, 2 * H, generator=g).to(bf).to(dev)
    nbytes = lib.mr_lstm_ws_bytes(1, T, N, H)
    res = {}
    for persist in (1, 0):
        out = torch.zeros(T, N, 2 * H, dtype=bf, device=dev)
        cbuf = torch.zeros(T, N, 2 * H, dtype=torch.float32, device=dev)
        gates = torch.zeros(T, N, 8 * H, dtype=bf, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if persist else None
        call("mr_lstm_fwd", 1, ptr(xproj), ptr(whh), ptr(out), ptr(cbuf), ptr(gates), T, N, H, ptr(ws),
             nbytes if persist else 0)
        torch.cuda.synchronize()
        if persist:
            print("fwd status", int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()))
        res[persist] = (out, cbuf, gates)
    for name, i in (("out", 0), ("cbuf", 1), ("gates", 2)):
        a, b = res[1][i].float(), res[0][i].float()
        print("fwd %-5s max|d| %.3e (max|ref| %.3e)" % (name, float((a - b).abs().max()), float(b.abs().max())))
    # backward from the SAME forward state (the per-step one)
    out, cbuf, gates0 = res[0]
    dres = {}
    for persist in (1, 0):
        gates = gates0.clone()
        dc = torch.zeros(N, 2 * H, dtype=torch.float32, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if persist else None
        call("mr_lstm_bwd", 1, ptr(dout), ptr(whh_t), ptr(cbuf), ptr(gates), ptr(dc), T, N, H, ptr(ws),
             nbytes if persist else 0)
        torch.cuda.synchronize()
        if persist:
            print("bwd status", int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()))
        dres[persist] = gates.float().view(T, N, 2, H, 4)
    a, b = dres[1], dres[0]
    print("bwd dgates max|d| %.3e (max|ref| %.3e)" % (float((a - b).abs().max()), float(b.abs().max())))
    for d in range(2):
        order = range(T - 1, -1, -1) if d == 0 else range(T)
        for si, t in enumerate(order):
            e = (a[t, :, d] - b[t, :, d]).abs()              # [N, H, 4]
            per_slice = [float(e[:, 64 * s:64 * s + 64].max()) for s in range(4)]
            per_gate = [float(e[..., q].max()) for q in range(4)]
            rows = e.amax(dim=(1, 2))
            if si < 4 or si == T - 1:
                print("  dir %d bwd-step %2d (t=%2d): max|d| per unit slice %s  per gate %s  worst row %d (%.2e)" %
                      (d, si, t, ["%.1e" % v for v in per_slice], ["%.1e" % v for v in per_gate],
                       int(rows.argmax()), float(rows.max())))


if __name__ == "__main__":
    main()
