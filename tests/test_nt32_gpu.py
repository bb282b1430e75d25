"""Ping-pong 32x32x16 NT kernel (csrc/igemm_nt32.h, mr_tuning.nt_m32; round 6) against the round-5 NT kernels and float64.

With integer-valued operands every product and partial sum is exact in f32, so the new kernel (another MFMA shape, another
summation order, a permuted accumulator layout) must agree BIT FOR BIT with the kernels it replaces -- output, fused bias + ReLU,
and the BatchNorm statistics of the epilogue; with random operands it must agree to bf16 rounding and match float64.
Reference of the operation: cuDNN / cuBLAS behind nn.Conv2d / nn.Linear (backbones/crnn.py:46-55, decoders/crnn.py:13-24)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd import _lib  # noqa: E402
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402
from megreader_amd.nn import functional as F  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def _rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _with(fields, fn):
    old = _lib.set_tuning(**fields)
    try:
        return fn()
    finally:
        _lib.set_tuning(**old)


# (M, N, K): whole tiles, ragged rows, ragged columns (N not a multiple of the tile), K not a multiple of 64, a single k-tile,
# an odd number of k-tiles
GEMMS = [(512, 512, 1024), (8448, 2048, 512), (700, 320, 576), (300, 200, 200), (256, 256, 64), (1000, 136, 192)]


@pytest.mark.parametrize("shape", [2, 3, 4, 5])
@pytest.mark.parametrize("M,N,K", GEMMS)
def test_dense_gemm_equals_round5_kernels(shape, M, N, K):
    mr.set_compute_dtype(BF)
    g = torch.Generator().manual_seed(M + N + K)
    ldc = (N + 7) // 8 * 8
    bias_i = torch.randint(-4, 5, (N,), generator=g).float().to(DEV)

    def run(A, B, bias, m32):
        def launch():
            C = torch.full((M, ldc), 7.0, device=DEV, dtype=BF)
            call("mr_gemm_nt", dtype_code(BF), ptr(A), K, ptr(B), K, ptr(C), ldc, ptr(bias), 1, M, N, K)
            return C
        return _with(dict(nt_m32=m32), launch)

    A = torch.randint(-3, 4, (M, K), generator=g).float().to(DEV, BF)
    B = torch.randint(-2, 3, (N, K), generator=g).float().to(DEV, BF)
    c0, c1 = run(A, B, bias_i, 0), run(A, B, bias_i, shape)
    assert torch.equal(c0, c1), (shape, M, N, K, float((c0.float() - c1.float()).abs().max()))
    A = torch.randn(M, K, generator=g).to(DEV, BF)
    B = torch.randn(N, K, generator=g).to(DEV, BF)
    bias = torch.randn(N, generator=g).to(DEV)
    c1 = run(A, B, bias, shape)
    ref = torch.relu(A.double().cpu() @ B.double().cpu().t() + bias.double().cpu())
    assert _rel_err(c1[:, :N], ref) < 1.6e-2
    if ldc != N:
        assert float((c1[:, N:].float() - 7.0).abs().max()) == 0.0   # pad columns untouched


# (batch, H, W, Cin, Cout): the CRNN layers the kernel serves by default (conv3 / conv5 at a smaller batch), rows that are
# not whole tiles, an image width that puts tile edges inside image rows (4 x 33)
CONVS = [(8, 8, 32, 256, 256), (9, 4, 33, 256, 512), (3, 7, 9, 128, 192), (16, 4, 33, 512, 512)]


@pytest.mark.parametrize("shape", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("N_,H,W,Cin,Cout", CONVS)
def test_conv_forward_dgrad_and_statistics_equal_round5_kernels(shape, N_, H, W, Cin, Cout):
    """3x3 / pad 1 convolution forward with the BatchNorm-statistics epilogue, and the input gradient, through the autograd
    wrapper (the path the models take); nt_m32 = 1 is the automatic choice (may or may not pick the new kernel for a shape)."""
    mr.set_compute_dtype(BF)
    g = torch.Generator().manual_seed(N_ + H + W + Cin)

    def run(x, w, dy, m32):
        def launch():
            xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wd = w.to(DEV).requires_grad_(True)
            y = F.conv2d(xd, wd, None, (1, 1), (1, 1), bn_stats=True)
            pre = getattr(y, "_mr_bn_sums", None)
            # f64 [copies][2][C]: which copy a workgroup adds to depends on the tile -> workgroup map; the totals must agree
            sums = pre.sums.flatten()[:16 * Cout].clone().view(8, 2, Cout).sum(0) if pre is not None else None
            y.backward(dy.to(DEV).contiguous(memory_format=torch.channels_last).to(y.dtype))
            return y.detach().clone(), xd.grad.detach().clone(), sums
        return _with(dict(nt_m32=m32, nt_big_min_k=64), launch)

    x = torch.randint(-3, 4, (N_, Cin, H, W), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin, 3, 3), generator=g).float()
    dy = torch.randint(-2, 3, (N_, Cout, H, W), generator=g).float()
    y0, dx0, s0 = run(x, w, dy, 0)
    y1, dx1, s1 = run(x, w, dy, shape)
    assert torch.equal(y0, y1), (shape, float((y0.float() - y1.float()).abs().max()))
    assert torch.equal(dx0, dx1), (shape, float((dx0.float() - dx1.float()).abs().max()))
    assert (s0 is None) == (s1 is None)
    if s0 is not None:
        assert torch.equal(s0, s1)
    x = torch.randn(N_, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    dy = torch.randn(N_, Cout, H, W, generator=g)
    y1, dx1, _ = run(x, w, dy, shape)
    xb, wb, dyb = x.to(BF).double(), w.to(BF).double(), dy.to(BF).double()
    xr = xb.clone().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wb, None, 1, 1)
    yr.backward(dyb)
    assert _rel_err(y1, yr.detach()) < 1.6e-2
    assert _rel_err(dx1, xr.grad) < 1.6e-2
