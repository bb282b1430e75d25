"""Small fixed workload for PMC passes of the all-taps wgrad kernel: CRNN conv2..conv5 weight gradients (N=256), 3 launches
each.  Usage: pmc_case_taps.py W8 GROUP"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megreader_amd import _lib  # noqa: E402
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402
from megreader_amd.nn import functional as F  # noqa: E402

lib = _lib.load()
F.set_tn_taps(1)
F.ensure_tn_taps_workspace("cuda")
lib.mr_set_tn_taps_w8(int(sys.argv[1]))
lib.mr_set_tn_taps_group(int(sys.argv[2]))
dt = dtype_code(torch.bfloat16)
N = 256
for H, W, C, K in [(8, 32, 128, 256), (8, 32, 256, 256), (4, 33, 256, 512), (4, 33, 512, 512)]:
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    dy = torch.randn(N, H, W, K, device="cuda").bfloat16()
    gw = torch.zeros(K, 3, 3, C, device="cuda")
    gb = torch.zeros(K, device="cuda")
    tab = torch.empty(N * H * W, 2, dtype=torch.int32, device="cuda")
    for i in range(3):
        call("mr_conv2d_wgrad_tab", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), N, H, W, C, C, K, K, 3, 3, 1, 1, 1, 1, 1, 1, H, W,
             ptr(tab), 1 if i == 0 else 0)
torch.cuda.synchronize()
