"""Small fixed workload for PMC passes: CRNN layer-5 conv forward (NT) and weight gradient (TN), 3 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402

dt = dtype_code(torch.bfloat16)
N, H, W, C, K, k, p = 256, 4, 33, 512, 512, 3, 1
x = torch.randn(N, H, W, C, device="cuda").bfloat16()
w = (torch.randn(K, k, k, C, device="cuda") * 0.05).bfloat16()
dy = torch.randn(N, H, W, K, device="cuda").bfloat16()
y = torch.empty(N, H, W, K, device="cuda", dtype=torch.bfloat16)
gw = torch.zeros(K, k, k, C, device="cuda")
gb = torch.zeros(K, device="cuda")
bias = torch.zeros(K, device="cuda")
for _ in range(3):
    call("mr_conv2d_fwd", dt, ptr(x), ptr(w), ptr(bias), ptr(y), 1, N, H, W, C, C, K, K, k, k, 1, 1, p, p, 1, 1, H, W)
    call("mr_conv2d_wgrad", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), N, H, W, C, C, K, K, k, k, 1, 1, p, p, 1, 1, H, W)
torch.cuda.synchronize()
