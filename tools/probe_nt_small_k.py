"""Memory-bound 1x1-conv-like GEMMs (small K): which 4-wave tile reaches the highest bandwidth?"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from megreader_amd import _lib
from megreader_amd._lib import call, ptr
from microbench_tn_taps import bench
lib = _lib.load()
for (M, N, K) in [(65536, 256, 64), (65536, 64, 256), (65536, 256, 128), (16384, 512, 128), (16384, 128, 512), (65536, 128, 64)]:
    A = torch.randn(M, K, device='cuda').bfloat16(); B = torch.randn(N, K, device='cuda').bfloat16()
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); bias = torch.zeros(N, device='cuda')
    line = "M=%d N=%d K=%d (%.1f MB):" % (M, N, K, (M * K + M * N + N * K) * 2e-6)
    for t in [(0, 0), (128, 128), (128, 64), (96, 128), (64, 128), (64, 64)]:
        lib.mr_force_nt_tile(*t)
        us = bench(lambda: call("mr_gemm_nt", 1, ptr(A), K, ptr(B), K, ptr(C), N, ptr(bias), 1, M, N, K), 30)
        line += "  %s %.1f" % ("auto" if t[0] == 0 else "%dx%d" % t, us)
    lib.mr_force_nt_tile(0, 0)
    print(line)
