"""All-taps weight-gradient kernel (csrc/tn_taps.hip) through the C ABI (mr_conv2d_wgrad_tab with mr_set_tn_taps(1)).

Operands are small integers, so every product and every partial sum is exact in bf16 / f32: the kernel's dW and dbias must
EQUAL torch's float64 convolution gradients bit for bit, whatever the order of its atomics.  Geometries: the CRNN 3x3
layers (reference backbones/crnn.py:44-55), a dilated ResNet layer (backbones/resnet_dilated.py), ragged channel counts,
ragged split boundaries, images smaller than one 64-position chunk.
"""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd import _lib  # noqa: E402
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402
from megreader_amd.nn import functional as F  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True, params=[0, 1], ids=["w4", "w8"])
def _taps_on(request):
    """Every test runs with the 4-wave (two workgroups per CU) and the 8-wave (one per CU) variant of the kernel."""
    old = F.set_tn_taps(1)
    oldp = _lib.set_tuning(tn_taps_min_p=0)      # the small test shapes must reach the all-taps kernel (default: P >= 10000 only)
    oldw = _lib.load().mr_set_tn_taps_w8(request.param)
    F.ensure_tn_taps_workspace(DEV)
    yield
    F.set_tn_taps(old)
    _lib.set_tuning(**oldp)
    _lib.load().mr_set_tn_taps_w8(oldw)
    _lib.load().mr_set_tn_splits(0)
    _lib.load().mr_set_tn_taps_group(0)


def _run(N, H, W, C, K, dil, splits=0, ldx=None, lddy=None, bias=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    ldx = ldx or C
    lddy = lddy or K
    x = torch.randint(-3, 4, (N, H, W, ldx), generator=g).float()
    dy = torch.randint(-3, 4, (N, H, W, lddy), generator=g).float()
    xd, dyd = x.to(DEV).bfloat16(), dy.to(DEV).bfloat16()
    gw = torch.zeros(K, 3, 3, C, device=DEV)
    gb = torch.zeros(K, device=DEV)
    tab = torch.empty(N * H * W, 2, dtype=torch.int32, device=DEV)
    _lib.load().mr_set_tn_splits(splits)
    assert _lib.load().mr_tn_taps_would_run(N, H, W, C, ldx, K, lddy, 3, 3, 1, 1, dil, dil, dil, dil, H, W) == 1
    call("mr_conv2d_wgrad_tab", dtype_code(torch.bfloat16), ptr(dyd), ptr(xd), ptr(gw), ptr(gb) if bias else 0, N, H,
         W, C, ldx, K, lddy, 3, 3, 1, 1, dil, dil, dil, dil, H, W, ptr(tab), 1)
    # second call re-uses the table (build = 0) and accumulates: result must be exactly twice the gradient
    call("mr_conv2d_wgrad_tab", dtype_code(torch.bfloat16), ptr(dyd), ptr(xd), ptr(gw), ptr(gb) if bias else 0, N, H,
         W, C, ldx, K, lddy, 3, 3, 1, 1, dil, dil, dil, dil, H, W, ptr(tab), 0)
    torch.cuda.synchronize()
    xr = x[..., :C].permute(0, 3, 1, 2).double()
    dyr = dy[..., :K].permute(0, 3, 1, 2).double()
    wref = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    TF.conv2d(xr, wref, None, 1, dil, dil).backward(dyr)
    ref = wref.grad.permute(0, 2, 3, 1)   # KRSC
    assert torch.equal(gw.cpu().double(), 2 * ref), "dW differs: max |err| %g" % float((gw.cpu().double() - 2 * ref).abs().max())
    if bias:
        assert torch.equal(gb.cpu().double(), 2 * dyr.sum((0, 2, 3)))
    else:
        assert float(gb.abs().max()) == 0


@pytest.mark.parametrize("N,H,W,C,K,dil", [
    (4, 8, 32, 128, 256, 1),     # conv2 of the CRNN at a small batch
    (6, 4, 33, 256, 512, 1),     # conv4: odd width
    (3, 4, 33, 512, 512, 1),     # conv5
    (2, 8, 32, 64, 64, 1),       # ResNet layer1 3x3
    (3, 6, 10, 64, 64, 2),       # dilated (resnet_dilated.py), image smaller than a chunk
    (5, 5, 7, 64, 72, 1),        # Cout not a multiple of the 64-row tile; IP rounded up to 8
    (2, 16, 16, 192, 40, 1),     # three Cin tiles, one partial Cout tile
])
def test_taps_wgrad_exact(N, H, W, C, K, dil):
    assert _lib.load().mr_set_tn_taps(1) == 1
    _run(N, H, W, C, K, dil)


@pytest.mark.parametrize("splits", [1, 2, 3, 7, 1000])
def test_taps_split_boundaries(splits):
    _run(7, 4, 33, 64, 64, 1, splits=splits)


@pytest.mark.parametrize("fin", [2, 1, 0])
@pytest.mark.parametrize("group", [1, 2, 3, 4, 8, 64])
@pytest.mark.parametrize("splits", [2, 5, 8])
def test_taps_group_reduction(group, splits, fin):
    """In-launch reduction of the split partials through slabs + tickets (TapArgs.grp): ragged last groups, groups larger
    than the split count, and back-to-back launches (the second launch of _run re-uses the tickets the first one reset)."""
    _lib.load().mr_set_tn_taps_group(group)
    # 2: every workgroup leaves its partial in its own slab, the finalize launch sums the splits (no tickets; `group` unused);
    # 1: finalize launch adds the group sums into dw; 0: the leaders' atomics
    old = _lib.load().mr_set_tn_taps_fin(fin)
    try:
        _run(9, 4, 33, 128, 128, 1, splits=splits)
    finally:
        _lib.load().mr_set_tn_taps_fin(old)


def test_taps_padded_row_strides_and_no_bias():
    _run(3, 8, 32, 64, 128, 1, ldx=80, lddy=136)
    _run(3, 8, 32, 64, 128, 1, bias=False)


def test_taps_full_size_matches_gemm_kernel():
    """CRNN conv5 at the benchmarked batch (N=256): all-taps kernel vs the 128x128 TN kernel on the same operands."""
    N, H, W, C, K = 256, 4, 33, 512, 512
    g = torch.Generator().manual_seed(1)
    x = torch.randint(-2, 3, (N, H, W, C), generator=g).to(DEV).bfloat16()
    dy = torch.randint(-2, 3, (N, H, W, K), generator=g).to(DEV).bfloat16()
    outs = []
    for mode in (1, 0):
        F.set_tn_taps(mode)
        gw = torch.zeros(K, 3, 3, C, device=DEV)
        gb = torch.zeros(K, device=DEV)
        tab = torch.empty(N * H * W, 2, dtype=torch.int32, device=DEV)
        call("mr_conv2d_wgrad_tab", dtype_code(torch.bfloat16), ptr(dy), ptr(x), ptr(gw), ptr(gb), N, H, W, C, C, K, K, 3,
             3, 1, 1, 1, 1, 1, 1, H, W, ptr(tab), 1)
        torch.cuda.synchronize()
        outs.append((gw, gb))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].abs().max()) > 0


def test_taps_through_autograd_conv():
    """Conv2dFn.backward reaches the kernel (bf16, 3x3, pad 1) and agrees with the default kernel."""
    mr.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    x = torch.randn(8, 128, 8, 32, device=DEV)
    w0 = torch.randn(256, 128, 3, 3, device=DEV) * 0.05
    b0 = torch.randn(256, device=DEV)
    grads = []
    for mode in (1, 0):
        F.set_tn_taps(mode)
        w = w0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True)
        y = F.conv2d(x, w, b, (1, 1), (1, 1), (1, 1), relu=False)
        (y.float() ** 2).sum().backward()
        grads.append((w.grad.clone(), b.grad.clone()))
    rel = float((grads[0][0] - grads[1][0]).abs().max() / grads[1][0].abs().max())
    relb = float((grads[0][1] - grads[1][1]).abs().max() / grads[1][1].abs().max())
    assert rel < 1e-5 and relb < 1e-5, (rel, relb)
    mr.set_compute_dtype(torch.bfloat16)
