#!/usr/bin/env python
"""1x1 convolutions (the K <= 2048 pointwise layers of the ResNet-50 workloads: profiles/r04_conv_shape_table_*.txt) through
mr_conv2d_fwd / mr_conv2d_dgrad with every 4-wave tile shape, the 2- and 4-buffer loops and the 8-wave big tiles, against the
automatic choice: which configuration wins per shape, and by how much.  bf16, back-to-back launches (warm L2 / MALL).
Usage: python tools/microbench_pointwise.py [--iters 20] [--set res50ppm|fpn|db|all]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402
from megreader_amd import _lib  # noqa: E402
from megreader_amd._lib import call, ptr  # noqa: E402

SHAPES = {   # (pixels M, Cout, Cin)
    "res50ppm": [(16384, 1024, 256), (16384, 256, 1024), (16384, 2048, 512), (16384, 512, 2048), (16384, 512, 128),
                 (16384, 128, 512), (65536, 256, 64), (65536, 64, 256), (16384, 512, 1024), (16384, 1024, 512)],
    "fpn": [(2048, 1024, 256), (2048, 256, 1024), (8192, 512, 128), (8192, 128, 512), (32768, 256, 64), (512, 2048, 512),
            (512, 512, 2048)],
    "db": [(3200, 1024, 256), (3200, 256, 1024), (12800, 512, 128), (12800, 128, 512), (51200, 256, 64), (800, 2048, 512),
           (800, 512, 2048)],
}
TILES = [(128, 128), (128, 64), (96, 128), (96, 64), (64, 128), (64, 64)]


def timeit(f, iters):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--set", default="all")
    a = ap.parse_args()
    mr.set_compute_dtype(torch.bfloat16)
    _lib.load()
    dt = 1
    sets = SHAPES.keys() if a.set == "all" else [a.set]
    base = _lib.get_tuning()
    configs = [("auto", {})]
    for deep in (0, 2):
        configs.append(("auto,deep=%d" % deep, dict(nt_deep=deep)))
    configs.append(("big_min_k=64", dict(nt_big_min_k=64)))
    for bm, bn in TILES:
        for deep in (0, 2):
            configs.append(("%dx%d,deep=%d" % (bm, bn, deep), dict(nt_force_bm=bm, nt_force_bn=bn, nt_deep=deep)))
    for s in sets:
        for M, Co, Ci in SHAPES[s]:
            H = W = 8
            N = M // 64
            x = torch.randn(N, H, W, Ci, device="cuda").bfloat16()
            w = (torch.randn(Co, Ci, device="cuda") * Ci ** -0.5).bfloat16()
            y = torch.empty(N, H, W, Co, device="cuda", dtype=torch.bfloat16)
            res = []
            for name, t in configs:
                _lib.set_tuning(**{k: base[k] for k in ("nt_deep", "nt_force_bm", "nt_force_bn", "nt_big_min_k")})
                _lib.set_tuning(**t)

                def f():
                    call("mr_conv2d_fwd", dt, ptr(x), ptr(w), 0, ptr(y), 0, N, H, W, Ci, Ci, Co, Co, 1, 1, 1, 1, 0, 0, 1, 1, H, W)
                res.append((timeit(f, a.iters), name))
            _lib.set_tuning(**{k: base[k] for k in ("nt_deep", "nt_force_bm", "nt_force_bn", "nt_big_min_k")})
            auto = res[0][0]
            best = min(res)
            gb = 2.0 * (M * Ci + M * Co + Co * Ci) / 1e9
            code = _lib.load().mr_nt_kernel_code(dt, M, Co, Ci, Ci)
            print("%-8s M=%6d N=%5d K=%5d | auto %7.2f us (%4.2f TB/s, kernel %d) | best %7.2f us %-18s (%.2fx) | %s" %
                  (s, M, Co, Ci, auto, gb / auto * 1e-3 * 1e3, code, best[0], best[1], auto / best[0],
                   "  ".join("%s=%.1f" % (n, t) for t, n in sorted(res)[:5])), flush=True)


if __name__ == "__main__":
    main()
