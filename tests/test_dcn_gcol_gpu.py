"""bf16 DCNv2 backward through a materialised gcol (csrc/dcn_fused.hip "materialised gcol", mr_tuning.dcn_gcol; round 6) against the
round-3 fused kernels (gcol in accumulators + CSR gather-GEMM) and the float64 autograd oracle, at the real layer shapes of the
published DB configuration.  Reference: assets/ops/dcn/src/deform_conv_cuda.cpp:611-675 (gcol GEMM, col2im, col2im_coord)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd import _lib  # noqa: E402
from megreader_amd.assets.ops.dcn import modulated_deform_conv  # noqa: E402
from oracle.dcn import modulated_deform_conv2d  # noqa: E402

DEV = "cuda"
LAYERS = [("layer2.0", 128, 160, 160, 2), ("layer2.1", 128, 80, 80, 1), ("layer3.0", 256, 80, 80, 2),
          ("layer3.1", 256, 40, 40, 1), ("layer4.0", 512, 40, 40, 2), ("layer4.1", 512, 20, 20, 1), ("c64", 64, 24, 20, 1)]


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_gcol_backward_vs_fused_kernels_and_oracle(layer):
    name, C, H, W, stride = layer
    N, Co, pad, dil = 2, C, 1, 1
    mr.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(C + H + stride)
    Ho = (H + 2 * pad - 3) // stride + 1
    Wo = (W + 2 * pad - 3) // stride + 1
    x = torch.randn(N, C, H, W, generator=g).bfloat16()
    off = torch.floor(torch.randn(N, 18, H, W, generator=g) * 1.5) + 0.25 + 0.5 * torch.rand(N, 18, H, W, generator=g)
    off[:, :, :2] += 1000.0                       # samples far outside the image: invalid, empty CSR rows
    msk = torch.sigmoid(torch.randn(N, 9, H, W, generator=g))
    msk[:, 4, ::3] = 0.0                          # exact zero masks: no CSR entry, still a mask gradient
    w = (torch.randn(Co, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5)
    gy = torch.randn(N, Co, Ho, Wo, generator=g).bfloat16()

    def run(gcol, col_fwd=1):
        old = _lib.set_tuning(dcn_gcol=gcol, dcn_col_fwd=col_fwd)
        try:
            xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            offd, mskd = off.to(DEV).requires_grad_(True), msk.to(DEV).requires_grad_(True)
            wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = modulated_deform_conv(xd, offd, mskd, wd, None, stride, pad, dil, 1, 1)
            y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
            torch.cuda.synchronize()
            return xd.grad.clone(), offd.grad.clone(), mskd.grad.clone(), wd.grad.clone()
        finally:
            _lib.set_tuning(**old)

    new, new2, old = run(1), run(1), run(0)
    resampled = run(1, 0)        # the backward samples the column matrix again instead of reading the forward's: the same matrix
    assert _rel(new[3], resampled[3]) < 1e-3
    # the coordinate pass has one writer per element and a fixed summation order: the same bits every run.  (dx sums each CSR list in
    # the order the fill pass happened to claim its slots -- f32 rounding differs run to run, as in the reference's atomic col2im,
    # quirk Q12; dW still uses f32 atomics.)
    assert torch.equal(new[1], new2[1]) and torch.equal(new[2], new2[2])
    assert _rel(new[0], new2[0]) < 1e-2
    xr = x.double().requires_grad_(True)
    offr, mskr = off.double().requires_grad_(True), msk.double().requires_grad_(True)
    wr = w.bfloat16().double().requires_grad_(True)
    yr = modulated_deform_conv2d(xr, offr, mskr, wr, None, stride, pad, dil)
    yr.backward(gy.double())
    ref = (xr.grad, offr.grad, mskr.grad, wr.grad)
    for nm, a, o, r in zip(("dx", "doffset", "dmask", "dw"), new, old, ref):
        e_new, e_old = _rel(a, r), _rel(o, r)
        print("%s %s: gcol path %.2e, fused kernels %.2e (vs float64)" % (name, nm, e_new, e_old))
        assert e_new < 3e-2, (nm, e_new)
        assert e_new < 2.0 * e_old + 4e-3, (nm, e_new, e_old)     # no worse than the kernels it replaces, up to bf16 rounding of gcol
