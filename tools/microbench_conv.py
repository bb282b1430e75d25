#!/usr/bin/env python
"""Per-layer timing of the conv kernels at the CRNN shapes (N=256, 32x128 input): fwd / dgrad / wgrad in TF/s.
Usage: python tools/microbench_conv.py [--dtype bf16|f32] [--iters 20] [--only fwd,dgrad,wgrad] [--layers 1,3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402

LAYERS = [  # H, W, Cin(phys), Cout, k, pad
    (32, 128, 8, 64, 3, 1), (16, 64, 64, 128, 3, 1), (8, 32, 128, 256, 3, 1), (8, 32, 256, 256, 3, 1),
    (4, 33, 256, 512, 3, 1), (4, 33, 512, 512, 3, 1), (2, 34, 512, 512, 2, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--layers", default="")
    ap.add_argument("--tile", default="", help="force NT tile, e.g. 96x128")
    ap.add_argument("--variant", type=int, default=2, help="NT kernel: 2 direct-to-LDS, 1 register staged")
    ap.add_argument("--tngroup", type=int, default=0, help="TN GEMM kernel split reduction: 0 auto, 1 atomics")
    ap.add_argument("--tnbuf", type=int, default=1, help="TN kernel staging through buffer resources (0/1)")
    ap.add_argument("--tnbig", type=int, default=0, help="big-tile TN kernel: 0 auto, -1 never, 1 always")
    ap.add_argument("--big", type=int, default=0, help="big-tile NT kernel: 0 auto, -1 never, 1 256x256, 2 288x256")
    ap.add_argument("--tnabl", type=int, default=0, help="timing-only ablation mask of the TN kernel")
    ap.add_argument("--p8", type=int, default=0, help="phased-schedule 256x256 kernel for the big-tile launches (0/1)")
    ap.add_argument("--tnmodel", type=int, default=0, help="1: dense-GEMM split model for conv wgrad too")
    ap.add_argument("--tnsplits", type=int, default=0, help="force the TN P-split count (0 = model)")
    ap.add_argument("--tune", default="", help="mr_tuning fields, e.g. nt_m32=2,nt_m32_opt=20")
    a = ap.parse_args()
    from megreader_amd import _lib
    _lib.load().mr_set_tn_model(a.tnmodel)
    _lib.load().mr_set_tn_splits(a.tnsplits)
    _lib.load().mr_set_nt_variant(a.variant)
    _lib.load().mr_set_nt_big(a.big)
    _lib.load().mr_set_nt_p8(a.p8)
    if a.tnabl or a.p8 > 1:   # wrong-result ablations: only in the `make ablation` build (MEGREADER_HIP_LIB=...abl.so)
        if not hasattr(_lib.load(), "mr_set_tn_abl"):
            raise SystemExit("--tnabl / --p8 > 1 need libmegreader_hip_abl.so: make -C megreader_amd/csrc ablation and "
                             "set MEGREADER_HIP_LIB to it")
        _lib.load().mr_set_tn_abl(a.tnabl)
    _lib.load().mr_set_tn_big(a.tnbig)
    _lib.load().mr_set_tn_buf(a.tnbuf)
    _lib.load().mr_set_tn_group(a.tngroup)
    if a.tile:
        bm, bn = [int(v) for v in a.tile.split("x")]
        assert _lib.load().mr_force_nt_tile(bm, bn) == 0
    if a.tune:
        _lib.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(","))})
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dt = dtype_code(dtype)
    dev = "cuda"
    N = a.batch
    sel = [int(x) for x in a.layers.split(",")] if a.layers else range(len(LAYERS))
    which = a.only.split(",")
    tot = {}
    for li in sel:
        H, W, C, K, k, p = LAYERS[li]
        Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
        x = torch.randn(N, H, W, C, device=dev).to(dtype)
        w = (torch.randn(K, k, k, C, device=dev) * 0.05).to(dtype)
        wt = (torch.randn(C, k, k, K, device=dev) * 0.05).to(dtype)
        dy = torch.randn(N, Ho, Wo, K, device=dev).to(dtype)
        y = torch.empty(N, Ho, Wo, K, device=dev, dtype=dtype)
        dx = torch.empty(N, H, W, C, device=dev, dtype=dtype)
        gw = torch.zeros(K, k, k, C, device=dev)
        gb = torch.zeros(K, device=dev)
        bias = torch.zeros(K, device=dev)
        tab = torch.empty(N * Ho * Wo, 2, dtype=torch.int32, device=dev)
        call("mr_conv2d_wgrad_tab", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), N, H, W, C, C, K, K, k, k, 1, 1, p, p, 1, 1,
             Ho, Wo, ptr(tab), 1)   # builds the row table (and runs once)
        gw.zero_(); gb.zero_()
        flops = 2.0 * N * Ho * Wo * K * k * k * C
        ops = {
            "fwd": lambda: call("mr_conv2d_fwd", dt, ptr(x), ptr(w), ptr(bias), ptr(y), 1, N, H, W, C, C, K, K, k, k,
                                1, 1, p, p, 1, 1, Ho, Wo),
            "dgrad": lambda: call("mr_conv2d_dgrad", dt, ptr(dy), ptr(wt), ptr(dx), N, H, W, C, C, K, K, k, k, 1, 1,
                                  p, p, 1, 1, Ho, Wo),
            "wgrad": lambda: call("mr_conv2d_wgrad", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), N, H, W, C, C, K, K, k,
                                  k, 1, 1, p, p, 1, 1, Ho, Wo),
            "wgradtab": lambda: call("mr_conv2d_wgrad_tab", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), N, H, W, C, C, K, K,
                                     k, k, 1, 1, p, p, 1, 1, Ho, Wo, ptr(tab), 0),
        }
        line = "L%d M=%7d N=%3d K=%4d " % (li, N * Ho * Wo, K, k * k * C)
        for name in which:
            if name == "dgrad" and li == 0:
                continue
            f = ops[name]
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / a.iters
            tot[name] = tot.get(name, 0.0) + us
            line += "| %s %7.1f us %6.1f TF " % (name, us, flops / us / 1e6)
        print(line)
    print("total us:", {k_: round(v, 1) for k_, v in tot.items()})


if __name__ == "__main__":
    main()
