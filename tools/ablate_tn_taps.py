#!/usr/bin/env python
"""Timing-only ablation ladder of the all-taps wgrad kernel (mr_set_tn_taps_abl): which ingredient costs what.
Needs the separate ablation build (wrong results by construction, not part of the product library):
    make -C megreader_amd/csrc ablation      # -> libmegreader_hip_abl.so (-DMR_ABLATION)"""
import os
import sys

import torch

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MEGREADER_HIP_LIB", os.path.join(_REPO, "megreader_amd", "csrc", "libmegreader_hip_abl.so"))
sys.path.insert(0, _REPO)
import megreader_amd as mr  # noqa: E402,F401
from megreader_amd import _lib  # noqa: E402
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402
from microbench_tn_taps import bench  # noqa: E402

CASES = [("crnn.conv5", 4, 33, 512, 512, 8), ("crnn.conv3", 8, 32, 256, 256, 32), ("crnn.conv4", 4, 33, 256, 512, 16)]
NAMES = {0: "full", 1: "no DMA", 2: "no x-fragment reads", 4: "no masks", 8: "no atomic epilogue", 16: "no barrier",
         32: "plain stores instead of atomics", 64: "a third of the atomics", 128: "atomics from 2 of 4 waves",
         256: "atomics from half the workgroups", 512: "group: no slab reads", 1024: "group: no final atomics",
         1536: "group: slab stores + ticket only", 2048: "DMA of L2-hot rows", 2056: "DMA of L2-hot rows, no epilogue",
         9: "no DMA, no epilogue", 6: "no x reads, no masks", 7: "no DMA/x reads/masks", 15: "MFMAs + dy reads + barrier", 31: "MFMAs + dy reads"}


def main():
    lib = _lib.load()
    dt = dtype_code(torch.bfloat16)
    N = 256
    lib.mr_set_tn_taps(1)
    lib.mr_set_tn_taps_w8(0)   # the ablation variants exist for the 4-wave kernel only
    lib.mr_set_tn_taps_group(4)
    from megreader_amd.nn import functional as F
    F.ensure_tn_taps_workspace('cuda')
    for name, H, W, C, K, splits in CASES:
        x = torch.randn(N, H, W, C, device="cuda").bfloat16()
        dy = torch.randn(N, H, W, K, device="cuda").bfloat16()
        gw = torch.zeros(K, 3, 3, C, device="cuda")
        gb = torch.zeros(K, device="cuda")
        tab = torch.empty(N * H * W, 2, dtype=torch.int32, device="cuda")
        flops = 2.0 * N * H * W * K * 9 * C
        lib.mr_set_tn_splits(splits)
        run = lambda b=0: call("mr_conv2d_wgrad_tab", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), N, H, W, C, C, K, K, 3, 3, 1, 1,
                               1, 1, 1, 1, H, W, ptr(tab), b)
        run(1)
        print("%s (splits %d)" % (name, splits))
        for m in (0, 8, 2048, 2056, 9):
            lib.mr_set_tn_taps_abl(m)
            us = bench(run, 20)
            print("   %-28s %7.1f us %6.0f TF/s" % (NAMES[m], us, flops / us * 1e-6), flush=True)
        lib.mr_set_tn_taps_abl(0)
    lib.mr_set_tn_splits(0)
    lib.mr_set_tn_taps(0)


if __name__ == "__main__":
    main()
