#!/usr/bin/env python
"""Dense TN GEMMs of the CRNN step (LSTM / Linear weight gradients, mr_gemm_tn): time per shape, with the automatic
P-split and a sweep of forced split counts (mr_set_tn_splits)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402
from megreader_amd._lib import call, load, ptr  # noqa: E402

SHAPES = [("L1 dW_ih", 8448, 2048, 512, 2048, 512), ("L1 dW_hh", 8192, 1024, 256, 2048, 512),
          ("L1 linear", 8448, 256, 512, 256, 512), ("L2 dW_ih", 8448, 2048, 256, 2048, 256),
          ("L2 linear", 8448, 40, 512, 40, 512)]


def timeit(f, iters=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    lib = load()
    sweep = [0, 2, 4, 8, 12, 16, 24, 32, 48, 64]
    for name, P, NA, NB, lda, ldb in SHAPES:
        A = torch.randn(P, lda, device="cuda").bfloat16()
        B = torch.randn(P, ldb, device="cuda").bfloat16()
        C = torch.zeros(NA, NB, device="cuda")
        cs = torch.zeros(NA, device="cuda")
        line = "%-10s P=%5d NA=%4d NB=%3d %5.1f GF |" % (name, P, NA, NB, 2e-9 * P * NA * NB)
        for s in sweep:
            lib.mr_set_tn_splits(s)
            g = torch.cuda.CUDAGraph()
            fn = lambda: call("mr_gemm_tn", 1, ptr(A), lda, ptr(B), ldb, ptr(C), NB, P, NA, NB, 0, ptr(cs))
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for _ in range(10):
                    fn()
            us = timeit(g.replay, 10) / 10
            line += " s=%d:%.1f" % (s, us)
        lib.mr_set_tn_splits(0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
