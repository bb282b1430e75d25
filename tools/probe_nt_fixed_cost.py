import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from megreader_amd import _lib
from megreader_amd._lib import call, ptr
from microbench_tn_taps import bench
lib = _lib.load()
for M in (32768, 65536, 262144):
    for K in (64, 128, 256, 576, 1152):
        N = 128
        A = torch.randn(M, K, device='cuda').bfloat16(); B = torch.randn(N, K, device='cuda').bfloat16()
        C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); bias = torch.zeros(N, device='cuda')
        us = bench(lambda: call("mr_gemm_nt", 1, ptr(A), K, ptr(B), K, ptr(C), N, ptr(bias), 1, M, N, K), 30)
        print("dense NT M=%d N=%d K=%d tile %d: %.1f us %.0f TF/s" % (M, N, K, lib.mr_nt_kernel_code(1, M, N, K, 0), us, 2.0*M*N*K/us*1e-6))
