"""DCNv2 (assets.ops.dcn mirror) on HIP vs the float64 autograd oracle (oracle/dcn.py), including the reference's
flat re-interpretation of a larger offset map for stride-2 blocks."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.assets.ops.dcn import ModulatedDeformConv, modulated_deform_conv  # noqa: E402
from oracle.dcn import modulated_deform_conv2d  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


CASES = [  # N, C, Co, H, W, stride, pad, dil, offset-map size (None = output grid), offset scale
    (2, 16, 24, 7, 9, 1, 1, 1, None, 2.0),
    (2, 32, 32, 12, 10, 2, 1, 1, (12, 10), 1.5),   # stride-2 conv, stride-1 offset map (quirk Q10)
    (1, 16, 16, 9, 8, 1, 2, 2, None, 1.0),          # dilated
    (3, 8, 8, 5, 6, 1, 1, 1, None, 4.0),            # large offsets: many invalid samples / border touches
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_forward_backward_vs_oracle(dtype, case):
    N, C, Co, H, W, stride, pad, dil, omap, oscale = case
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(C + H)
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    oh, ow = omap if omap else (Ho, Wo)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    # keep sample points away from exact integers (kink of the bilinear kernel; bf16 rounding could flip floor())
    off = (torch.randn(N, 18, oh, ow, generator=g) * oscale)
    off = torch.floor(off) + 0.25 + 0.5 * torch.rand(off.shape, generator=g)
    msk = torch.rand(N, 9, oh, ow, generator=g)
    w = (torch.randn(Co, C, 3, 3, generator=g) * 0.2)
    b = torch.randn(Co, generator=g)
    gy = torch.randn(N, Co, Ho, Wo, generator=g).to(dtype)

    xr = x.double().requires_grad_(True)
    offr = off.double().requires_grad_(True)
    mskr = msk.double().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = modulated_deform_conv2d(xr, offr, mskr, wr, br, stride, pad, dil)
    yr.backward(gy.double())

    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    offd = off.to(DEV).requires_grad_(True)
    mskd = msk.to(DEV).requires_grad_(True)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    y = modulated_deform_conv(xd, offd, mskd, wd, bd, stride, pad, dil, 1, 1)
    assert y.shape == yr.shape
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert _rel(y, yr) < tol
    y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    gtol = 1e-4 if dtype == torch.float32 else 3e-2
    assert _rel(xd.grad, xr.grad) < gtol
    assert _rel(wd.grad, wr.grad) < gtol
    assert _rel(bd.grad, br.grad) < gtol
    assert _rel(offd.grad, offr.grad) < gtol
    assert _rel(mskd.grad, mskr.grad) < gtol
    if omap:  # entries outside the flat [18,Ho,Wo] window get exactly zero gradient, as in the reference
        flat = offd.grad.reshape(N, -1)
        assert float(flat[:, 18 * Ho * Wo:].abs().max()) == 0.0


def test_zero_offset_module_equals_conv():
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    m = ModulatedDeformConv(16, 24, 3, stride=1, padding=1, bias=True).to(DEV)
    x = torch.randn(2, 16, 6, 7, device=DEV)
    off = torch.zeros(2, 18, 6, 7, device=DEV)
    msk = torch.ones(2, 9, 6, 7, device=DEV)
    y = m(x, off, msk)
    ref = TF.conv2d(x.cpu().double(), m.weight.detach().cpu().double(), m.bias.detach().cpu().double(), 1, 1)
    assert _rel(y, ref) < 2e-5


def test_cpu_raises_like_reference():
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(torch.zeros(1, 8, 4, 4), torch.zeros(1, 18, 4, 4), torch.ones(1, 9, 4, 4),
                              torch.zeros(8, 8, 3, 3))
