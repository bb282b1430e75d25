#!/usr/bin/env python
"""Weight-gradient kernels at the CRNN 3x3 layers (N=256): 128x128 TN GEMM kernel vs the all-taps kernel
(csrc/tn_taps.hip), with a sweep over the split count.  TFLOP/s are algorithmic (2*P*Cout*9*Cin).
Usage: python tools/microbench_tn_taps.py [--iters 20] [--sweep 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402,F401
from megreader_amd import _lib  # noqa: E402
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402

LAYERS = [  # name, N-scale, H, W, Cin, Cout
    ("crnn.conv2", 8, 32, 128, 256), ("crnn.conv3", 8, 32, 256, 256), ("crnn.conv4", 4, 33, 256, 512),
    ("crnn.conv5", 4, 33, 512, 512), ("res.l1", 8, 32, 64, 64), ("res.l2", 4, 16, 128, 128),
    ("res.l3", 2, 8, 256, 256)]


def bench(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--sweep", type=int, default=1)
    ap.add_argument("--layers", default="")
    ap.add_argument("--fin", type=int, default=0, help="1: finalize launch instead of the leaders' atomics")
    ap.add_argument("--w8", type=int, default=0, help="1: 8-wave workgroup variant")
    ap.add_argument("--group", type=int, default=0, help="group size of the in-launch split reduction (0 auto, 1 atomics)")
    a = ap.parse_args()
    lib = _lib.load()
    dt = dtype_code(torch.bfloat16)
    N = a.batch
    from megreader_amd.nn import functional as F
    F.ensure_tn_taps_workspace("cuda")
    lib.mr_set_tn_taps_group(a.group)
    lib.mr_set_tn_taps_w8(a.w8)
    lib.mr_set_tn_taps_fin(a.fin)
    total = {0: 0.0, 1: 0.0}
    for name, H, W, C, K in LAYERS:
        if a.layers and name not in a.layers.split(","):
            continue
        x = torch.randn(N, H, W, C, device="cuda").bfloat16()
        dy = torch.randn(N, H, W, K, device="cuda").bfloat16()
        gw = torch.zeros(K, 3, 3, C, device="cuda")
        gb = torch.zeros(K, device="cuda")
        flops = 2.0 * N * H * W * K * 9 * C
        line = "%-11s P=%6d K=%3d C=%3d :" % (name, N * H * W, K, C)
        ref = None
        for mode in (0, 1):
            lib.mr_set_tn_taps(mode)
            lib.mr_set_tn_splits(0)
            tab = torch.empty(N * H * W, 2, dtype=torch.int32, device="cuda")
            run = lambda b=0: call("mr_conv2d_wgrad_tab", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), N, H, W, C, C, K, K, 3, 3,
                                   1, 1, 1, 1, 1, 1, H, W, ptr(tab), b)
            gw.zero_()
            run(1)
            torch.cuda.synchronize()
            if ref is None:
                ref = gw.clone()
            else:
                line += " relerr %.1e" % float((gw - ref).abs().max() / ref.abs().max())
            us = bench(run, a.iters)
            total[mode] += us if name.startswith("crnn") else 0.0
            line += "  %s %7.1f us %6.0f TF/s" % ("taps" if mode else "gemm", us, flops / us * 1e-6)
            if mode == 1 and a.sweep:
                best = (1e30, 0)
                sw = ""
                for s in (2, 4, 6, 8, 12, 16, 24, 32, 48, 64):
                    lib.mr_set_tn_splits(s)
                    t = bench(run, max(5, a.iters // 2))
                    sw += " %d:%.0f" % (s, t)
                    best = min(best, (t, s))
                lib.mr_set_tn_splits(0)
                line += "  | splits sweep (us)" + sw + "  best %d" % best[1]
        print(line, flush=True)
    print("CRNN conv2..5 wgrad per step: gemm %.0f us, taps %.0f us" % (total[0], total[1]))
    lib.mr_set_tn_taps(0)


if __name__ == "__main__":
    main()
