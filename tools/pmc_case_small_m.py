"""Small-M NT launches for PMC / trace passes: layer4 conv2 (3x3, 512 -> 512) and conv3 (1x1, 512 -> 2048) of ResNet-50 at the
FPN-attention workload's geometry (batch 32, 2 x 8 pixels: M = 512 rows), forward and dgrad, 5 launches each.
usage: python tools/pmc_case_small_m.py <nt_deep mode: 0 | 1 | 2>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megreader_amd._lib import call, dtype_code, load, ptr  # noqa: E402

load().mr_set_nt_deep(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
dt = dtype_code(torch.bfloat16)
N, H, W = 32, 2, 8
for (C, K, k, p) in [(512, 512, 3, 1), (512, 2048, 1, 0), (2048, 512, 1, 0)]:
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    w = (torch.randn(K, k, k, C, device="cuda") * 0.05).bfloat16()
    wt = (torch.randn(C, k, k, K, device="cuda") * 0.05).bfloat16()
    dy = torch.randn(N, H, W, K, device="cuda").bfloat16()
    y = torch.empty(N, H, W, K, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        call("mr_conv2d_fwd", dt, ptr(x), ptr(w), 0, ptr(y), 0, N, H, W, C, C, K, K, k, k, 1, 1, p, p, 1, 1, H, W)
        call("mr_conv2d_dgrad", dt, ptr(dy), ptr(wt), ptr(dx), N, H, W, C, C, K, K, k, k, 1, 1, p, p, 1, 1, H, W)
torch.cuda.synchronize()
