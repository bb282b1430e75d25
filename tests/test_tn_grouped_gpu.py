"""Deferred, grouped weight-gradient launches (mr_tn_defer / mr_tn_flush, igemm_tn_glds_grouped_kernel) through the C ABI.

The weight gradients of the small layers (reference backbones/resnet.py:113-181 1x1 / strided convolutions, decoders/crnn.py:8-24
LSTM / Linear layers) are recorded and launched several problems per launch.  Operands are small integers, so every product and
partial sum is exact in bf16 / f32: a grouped launch must EQUAL float64 arithmetic bit for bit, whatever its split / atomic
order -- and therefore equal the immediate launches exactly.
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
BF = dtype_code(torch.bfloat16)


def _ints(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=g).float()


def _dense_problem(g, P, NA, NB, lda=None, ldb=None, perm_h=0, colsum=False):
    lda = lda or (NA + 7) // 8 * 8
    ldb = ldb or NB
    A = _ints((P, lda), g)
    if lda > NA:
        A[:, NA:] = 0                      # padding columns of A feed output rows that are never stored
    B = _ints((P, ldb), g)
    ref = A[:, :NA].double().t() @ B[:, :NB].double()
    if perm_h:                               # gate-interleaved rows back to gate-major order (include/megreader_hip.h: mr_gemm_tn)
        h4 = 4 * perm_h
        rows = torch.arange(NA)
        blk, rin = rows // h4, rows % h4
        dst = blk * h4 + (rin % 4) * perm_h + rin // 4
        out = torch.empty_like(ref)
        out[dst] = ref
        cs = torch.empty(NA, dtype=torch.float64)
        cs[dst] = A[:, :NA].double().sum(0)
        ref = out
    else:
        cs = A[:, :NA].double().sum(0)
    return {"A": A.to(DEV).bfloat16(), "B": B.to(DEV).bfloat16(), "P": P, "NA": NA, "NB": NB, "lda": lda, "ldb": ldb,
            "perm_h": perm_h, "C": torch.zeros(NA, NB, device=DEV), "cs": torch.zeros(NA, device=DEV) if colsum else None,
            "ref": ref, "cs_ref": cs}


def _launch_dense(pr):
    call("mr_gemm_tn", BF, ptr(pr["A"]), pr["lda"], ptr(pr["B"]), pr["ldb"], ptr(pr["C"]), pr["NB"], pr["P"], pr["NA"], pr["NB"],
         pr["perm_h"], ptr(pr["cs"]))


def _check_dense(pr, times=1):
    got = pr["C"].cpu().double()
    assert torch.equal(got, times * pr["ref"]), "max |err| %g" % float((got - times * pr["ref"]).abs().max())
    if pr["cs"] is not None:
        assert torch.equal(pr["cs"].cpu().double(), times * pr["cs_ref"])


DENSE_SHAPES = [
    # (P, NA, NB, lda, perm_h, colsum)  -- the CRNN head at a small batch: W_ih per direction, W_hh, the two Linear layers
    (33 * 8, 1024, 512, 2048, 256, True),
    (32 * 8, 1024, 256, 2048, 256, False),
    (33 * 8, 256, 512, None, 0, True),
    (33 * 8, 38, 512, 40, 0, True),
    (100, 130, 72, 136, 0, True),            # ragged everything: partial tiles in both directions, P not a multiple of 64
    (64, 8, 8, None, 0, False),
    (5000, 128, 128, None, 0, True),         # long reduction, one tile
]


@pytest.mark.parametrize("count", [2, 7, 12, 13, 20])
def test_grouped_dense_equals_float64(count):
    lib = _lib.load()
    g = torch.Generator().manual_seed(count)
    probs = [_dense_problem(g, P, NA, NB, lda=lda, perm_h=ph, colsum=cs)
             for (P, NA, NB, lda, ph, cs) in (DENSE_SHAPES * 3)[:count]]
    assert lib.mr_tn_pending() == 0
    old = lib.mr_tn_defer(1)
    for pr in probs:
        _launch_dense(pr)
    lib.mr_tn_defer(old)
    assert lib.mr_tn_pending() == count           # recorded, nothing launched ...
    torch.cuda.synchronize()
    assert all(float(pr["C"].abs().max()) == 0 for pr in probs)
    call("mr_tn_flush")
    assert lib.mr_tn_pending() == 0
    torch.cuda.synchronize()
    for pr in probs:
        _check_dense(pr)
    # a second round accumulates (the kernels add into C): exactly twice the gradient; this time mixed with immediate launches
    lib.mr_tn_defer(1)
    for pr in probs[::2]:
        _launch_dense(pr)
    lib.mr_tn_defer(0)
    for pr in probs[1::2]:
        _launch_dense(pr)                         # immediate (deferral off)
    call("mr_tn_flush")
    torch.cuda.synchronize()
    for pr in probs:
        _check_dense(pr, times=2)


@pytest.mark.parametrize("defer", [0, 1])
def test_gemm_tn2_adds_the_column_sums_to_both_destinations(defer):
    lib = _lib.load()
    g = torch.Generator().manual_seed(77)
    probs = [_dense_problem(g, 33 * 8, 1024, 512, lda=2048, perm_h=256, colsum=True) for _ in range(3)]
    second = [torch.full((1024,), 5.0, device=DEV) for _ in probs]
    lib.mr_tn_defer(defer)
    for pr, c2 in zip(probs, second):
        call("mr_gemm_tn2", BF, ptr(pr["A"]), pr["lda"], ptr(pr["B"]), pr["ldb"], ptr(pr["C"]), pr["NB"], pr["P"], pr["NA"],
             pr["NB"], pr["perm_h"], ptr(pr["cs"]), ptr(c2))
    lib.mr_tn_defer(0)
    call("mr_tn_flush")
    torch.cuda.synchronize()
    for pr, c2 in zip(probs, second):
        _check_dense(pr)
        assert torch.equal(c2.cpu().double(), 5.0 + pr["cs_ref"])


def test_defer_is_ignored_when_switched_off_or_f32():
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    pr = _dense_problem(g, 200, 64, 64)
    old_t = _lib.set_tuning(tn_defer=0)
    try:
        lib.mr_tn_defer(1)
        _launch_dense(pr)
        lib.mr_tn_defer(0)
        assert lib.mr_tn_pending() == 0
        torch.cuda.synchronize()
        _check_dense(pr)
    finally:
        _lib.set_tuning(**old_t)
    # float32 problems run on another kernel: never recorded
    A = _ints((100, 32), g).to(DEV)
    B = _ints((100, 16), g).to(DEV)
    C = torch.zeros(32, 16, device=DEV)
    lib.mr_tn_defer(1)
    call("mr_gemm_tn", dtype_code(torch.float32), ptr(A), 32, ptr(B), 16, ptr(C), 16, 100, 32, 16, 0, 0)
    lib.mr_tn_defer(0)
    assert lib.mr_tn_pending() == 0
    torch.cuda.synchronize()
    assert torch.equal(C.cpu().double(), A.cpu().double().t() @ B.cpu().double())


CONV_SHAPES = [
    # (N, H, W, C, K, R, S, stride, pad, dil): layers that stay on the 128x128 TN GEMM kernel
    (4, 8, 8, 256, 64, 1, 1, 1, 0, 1),       # ResNet bottleneck 1x1
    (4, 8, 8, 64, 256, 1, 1, 1, 0, 1),
    (3, 9, 9, 128, 128, 3, 3, 2, 1, 1),      # strided 3x3
    (4, 8, 8, 256, 512, 1, 1, 2, 0, 1),      # downsample branch
    (5, 2, 34, 512, 512, 2, 2, 1, 0, 1),     # conv6 of the CRNN (2x2, no padding)
    (2, 16, 64, 64, 128, 3, 3, 1, 1, 1),     # conv1 of the CRNN (W = 64: not eligible for the all-taps kernel)
    (2, 10, 10, 64, 27, 3, 3, 1, 1, 1),      # 27-channel DCN offset convolution stored with 32 channels
]


def _conv_problem(g, N, H, W, C, K, R, S, st, pad, dil):
    Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // st + 1, (W + 2 * pad - dil * (S - 1) - 1) // st + 1
    lddy = (K + 7) // 8 * 8
    x = _ints((N, H, W, C), g)
    dy = _ints((N, Ho, Wo, lddy), g)
    dy[..., K:] = 0
    wref = torch.zeros(K, C, R, S, dtype=torch.float64, requires_grad=True)
    TF.conv2d(x.permute(0, 3, 1, 2).double(), wref, None, st, pad, dil).backward(dy[..., :K].permute(0, 3, 1, 2).double())
    return {"x": x.to(DEV).bfloat16(), "dy": dy.to(DEV).bfloat16(), "gw": torch.zeros(K, R, S, C, device=DEV),
            "gb": torch.zeros(K, device=DEV), "tab": torch.empty(N * Ho * Wo, 2, dtype=torch.int32, device=DEV),
            "geom": (N, H, W, C, C, K, lddy, R, S, st, st, pad, pad, dil, dil, Ho, Wo),
            "ref": wref.grad.permute(0, 2, 3, 1), "b_ref": dy[..., :K].double().sum((0, 1, 2))}


def _launch_conv(pr, build):
    call("mr_conv2d_wgrad_tab", BF, ptr(pr["dy"]), ptr(pr["x"]), ptr(pr["gw"]), ptr(pr["gb"]), *pr["geom"], ptr(pr["tab"]), build)


@pytest.mark.parametrize("count", [3, 7, 14])
def test_grouped_conv_wgrad_equals_float64(count):
    lib = _lib.load()
    old_p = _lib.set_tuning(tn_taps_min_p=1 << 30)      # keep every shape off the all-taps kernel
    F._ROWTABS.clear()
    try:
        g = torch.Generator().manual_seed(100 + count)
        probs = [_conv_problem(g, *shape) for shape in (CONV_SHAPES * 2)[:count]]
        lib.mr_tn_defer(1)
        for pr in probs:
            _launch_conv(pr, 1)                         # the row-table kernel runs at once, the GEMM is recorded
        lib.mr_tn_defer(0)
        assert lib.mr_tn_pending() == count
        call("mr_tn_flush")
        for pr in probs:
            _launch_conv(pr, 0)                         # immediate second round on the same tables
        torch.cuda.synchronize()
        for pr in probs:
            got = pr["gw"].cpu().double()
            assert torch.equal(got, 2 * pr["ref"]), "dW: max |err| %g" % float((got - 2 * pr["ref"]).abs().max())
            assert torch.equal(pr["gb"].cpu().double(), 2 * pr["b_ref"])
    finally:
        _lib.set_tuning(**old_p)
        F._ROWTABS.clear()


def _crnn_head(T, N, seed):
    from megreader_amd.decoders.crnn import BidirectionalLSTM
    torch.manual_seed(seed)
    net = torch.nn.Sequential(BidirectionalLSTM(512, 256, 256), BidirectionalLSTM(256, 256, 38)).to(DEV)
    x = torch.randn(T, N, 512, device=DEV).bfloat16()
    return net, x


def test_bilstm_head_backward_grouped_vs_immediate():
    """The CRNN head (2 x BiLSTM + Linear, decoders/crnn.py:8-24) under FusedAdam sinks: the grouped launch (four weight
    gradients per layer + the Linear's in one launch, W_ih straight into its sinks) against the separate launches."""
    from megreader_amd.optim import FusedAdam
    mr.set_compute_dtype(torch.bfloat16)
    grads = {}
    for mode in (0, 1):
        old_t = _lib.set_tuning(tn_defer=mode)
        try:
            net, x = _crnn_head(9, 32, seed=3)
            opt = FusedAdam(net.parameters(), lr=0.0)
            opt.zero_grad()
            xin = x.clone().requires_grad_(True)
            y = net(xin)
            w = torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y).to(y.dtype)
            (y * w).sum().backward()
            torch.cuda.synchronize()
            assert _lib.load().mr_tn_pending() == 0
            grads[mode] = {k: p.grad.detach().float().cpu().clone() for k, p in net.named_parameters()}
            grads[mode]["x"] = xin.grad.float().cpu()
        finally:
            _lib.set_tuning(**old_t)
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale, (k, float((a - b).abs().max()) / scale)
