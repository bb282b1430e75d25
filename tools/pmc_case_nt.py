"""Fixed workload for PMC passes of the NT convolution kernels: CRNN conv3 (65536 x 256 x 2304) and conv5 (33792 x 512 x 4608) forward,
5 launches each, under the mr_tuning fields given as name=value arguments (e.g. nt_m32=2 nt_m32_opt=20)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megreader_amd import _lib  # noqa: E402
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402

fields = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:])}
if fields:
    _lib.set_tuning(**fields)
dt = dtype_code(torch.bfloat16)
for (N, H, W, C, K) in ((256, 8, 32, 256, 256), (256, 4, 33, 512, 512)):
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    w = (torch.randn(K, 3, 3, C, device="cuda") * 0.05).bfloat16()
    y = torch.empty(N, H, W, K, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(K, device="cuda")
    for _ in range(5):
        call("mr_conv2d_fwd", dt, ptr(x), ptr(w), ptr(bias), ptr(y), 1, N, H, W, C, C, K, K, 3, 3, 1, 1, 1, 1, 1, 1, H, W)
torch.cuda.synchronize()
