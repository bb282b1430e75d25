"""Fixed cost of an NT launch: dense GEMMs of the CRNN conv3 / conv5 output shapes at K = 64 ... 4608 under a list of mr_tuning
settings -- the slope over K is the steady-state k-tile time, the intercept the launch + prologue + epilogue cost.
usage: python tools/probe_nt_fixed_cost.py "nt_m32=0" "nt_m32=2" ..."""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from megreader_amd import _lib
from megreader_amd._lib import call, ptr


def bench(f, iters=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


for cfg in sys.argv[1:] or ["nt_m32=0"]:
    old = _lib.set_tuning(nt_big_min_k=32, **{k: int(v) for k, v in (kv.split("=") for kv in cfg.split(","))})
    for (M, N) in ((65536, 256), (33792, 512)):
        line = "%-28s M=%6d N=%3d |" % (cfg, M, N)
        for K in (64, 128, 256, 512, 1152, 2304, 4608):
            A = torch.randn(M, K, device='cuda').bfloat16(); B = torch.randn(N, K, device='cuda').bfloat16()
            C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); bias = torch.zeros(N, device='cuda')
            us = bench(lambda: call("mr_gemm_nt", 1, ptr(A), K, ptr(B), K, ptr(C), N, ptr(bias), 1, M, N, K))
            line += " K=%d %.1f" % (K, us)
        print(line, flush=True)
    _lib.set_tuning(**old)
x = torch.empty(65536, 256, device='cuda', dtype=torch.bfloat16)
y = torch.empty_like(x)
print("fill 33.5 MB: %.1f us; copy 33.5 MB: %.1f us" % (bench(lambda: x.fill_(1.0)), bench(lambda: y.copy_(x))))
