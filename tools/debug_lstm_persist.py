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
    dbg = torch.zeros(5, T, N, 2 * H, dtype=torch.float32, device=dev)
    for persist in (1, 0):
        gates = gates0.clone()
        lib.mr_lstm_debug_buffer(dbg.data_ptr() if persist else 0)
        dc = torch.zeros(N, 2 * H, dtype=torch.float32, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if persist else None
        call("mr_lstm_bwd", 1, ptr(dout), ptr(whh_t), ptr(cbuf), ptr(gates), ptr(dc), T, N, H, ptr(ws),
             nbytes if persist else 0)
        torch.cuda.synchronize()
        if persist:
            print("bwd status", int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item()))
        dres[persist] = gates.float().view(T, N, 2, H, 4)
    lib.mr_lstm_debug_buffer(0)
    # which source slices made it into the recurrent term?  expected partial of slice g at forward time t (dir 0):
    # dgates_persist[t+1][:, dir, slice g units, :] (as the kernel itself produced them, bf16) @ W_hh[those cols, :]
    dgp = dres[1]                                                # [T, N, 2, H, 4] f32 (bf16 values)
    for d in range(2):
        t = T - 2 if d == 0 else 1                               # second backward step of that direction
        tn = t + 1 if d == 0 else t - 1
        W = whh[d].float()                                       # [4H, H] gate-interleaved rows
        parts = []
        for gsl in range(4):
            cols = slice(256 * gsl, 256 * gsl + 256)
            parts.append(dgp[tn, :, d].reshape(N, 4 * H)[:, cols] @ W[cols, :])      # [N, H]
        got = dbg[0, t, :, d * H:(d + 1) * H]
        comps = [dbg[1 + c, t, :, d * H:(d + 1) * H] for c in range(4)]
        full = sum(parts)
        print("dir %d t=%d: recurrent term max|got - full| %.3e  (max|full| %.3e)" %
              (d, t, float((got - full).abs().max()), float(full.abs().max())))
        for dest in range(4):
            u = slice(64 * dest, 64 * dest + 64)
            errs = ["%.2e" % float((got[:, u] - p_[:, u]).abs().max()) for p_ in parts]
            print("   dest slice %d: |got - single source g| %s ; |got - full| %.2e ; |got| %.2e" %
                  (dest, errs, float((got[:, u] - full[:, u]).abs().max()), float(got[:, u].abs().max())))
            for ci, cname in enumerate(("own", "slab0", "slab1", "slab2")):
                e2 = ["%.2e" % float((comps[ci][:, u] - p_[:, u]).abs().max()) for p_ in parts]
                print("        component %-5s vs source g: %s   |comp| %.2e" %
                      (cname, e2, float(comps[ci][:, u].abs().max())))
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
