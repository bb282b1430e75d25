"""Per-kernel parity of the HIP library (through the C ABI / autograd wrappers) against torch CPU references.

fp32 mode uses the exact-f32 MFMA path: tolerances are f32 round-off class.  bf16 mode rounds operands to bf16,
so the reference is computed on bf16-rounded inputs and compared with a bf16-output tolerance.
"""
import math

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402
from megreader_amd.nn import functional as F  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def _tol(dtype, k=1):
    return (2e-5 * math.sqrt(k) + 1e-6) if dtype == torch.float32 else (1.6e-2)


def _rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (8448, 2048, 512), (200, 38, 512), (64, 1024, 256),
                                   (1000, 136, 72), (33, 512, 2304)])
def test_gemm_nt(dtype, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)  # asymmetric, M != N: catches transposed C writes
    bias = torch.randn(N, generator=g)
    Ad, Bd = A.to(DEV, dtype), B.to(DEV, dtype)
    ldc = (N + 7) // 8 * 8
    C = torch.full((M, ldc), 7.0, device=DEV, dtype=dtype)
    call("mr_gemm_nt", dtype_code(dtype), ptr(Ad), K, ptr(Bd), K, ptr(C), ldc, ptr(bias.to(DEV)), 1, M, N, K)
    ref = torch.relu(Ad.double().cpu() @ Bd.double().cpu().t() + bias.double())
    assert _rel_err(C[:, :N], ref) < _tol(dtype, K)
    if ldc != N:
        assert float((C[:, N:].float() - 7.0).abs().max()) == 0.0  # pad columns untouched


@pytest.mark.parametrize("case", [("gemm", 512, 512, 4608), ("gemm", 300, 200, 2048), ("gemm", 64, 1024, 1056),
                                  ("conv", 2, 16, 16, 512, 256, 3), ("conv", 3, 7, 9, 256, 192, 3), ("conv", 2, 20, 20, 1024, 256, 1)])
def test_nt_split_reduction_is_exact_and_repeatable(case):
    """mr_tuning.nt_ksplit (round 5): launches of a few tiles with a long k-loop are cut along the reduction; the partial
    tiles meet in f32 slabs and are added in split order.  With integer-valued operands every product and partial sum is exact,
    so split and unsplit launches must agree BIT FOR BIT (output, fused bias + ReLU, BatchNorm statistics of the epilogue); with
    random operands two split launches must agree bit for bit with each other (fixed summation order) and with the unsplit one
    to bf16 rounding."""
    from megreader_amd import _lib
    _lib.ensure_tn_workspace(DEV)
    mr.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(len(case) + case[1] + case[3])

    def run(ints):
        if case[0] == "gemm":
            _, M, N, K = case
            A = (torch.randint(-3, 4, (M, K), generator=g).float() if ints else torch.randn(M, K, generator=g)).to(DEV, torch.bfloat16)
            B = (torch.randint(-2, 3, (N, K), generator=g).float() if ints else torch.randn(N, K, generator=g)).to(DEV, torch.bfloat16)
            bias = torch.randint(-4, 5, (N,), generator=g).float().to(DEV)
            ldc = (N + 7) // 8 * 8

            def launch():
                C = torch.zeros((M, ldc), device=DEV, dtype=torch.bfloat16)
                call("mr_gemm_nt", dtype_code(torch.bfloat16), ptr(A), K, ptr(B), K, ptr(C), ldc, ptr(bias), 1, M, N, K)
                return (C,)
        else:
            _, N_, H, W, Cin, Cout, R = case
            x = (torch.randint(-3, 4, (N_, Cin, H, W), generator=g).float() if ints else torch.randn(N_, Cin, H, W, generator=g))
            w = (torch.randint(-2, 3, (Cout, Cin, R, R), generator=g).float() if ints else
                 torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5)
            xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
            wd = w.to(DEV)

            def launch():
                y = F.conv2d(xd, wd, None, (1, 1), (R // 2, R // 2), bn_stats=True)
                pre = getattr(y, "_mr_bn_sums", None)     # the BatchNorm statistics of the epilogue (arena slice: clone now)
                return (y,) + ((pre.sums.clone(),) if pre is not None else ())
        outs = {}
        for mode in (0, 1, 1, 4):
            old = _lib.set_tuning(nt_ksplit=mode)
            try:
                outs.setdefault(mode, []).append([t.detach().clone() for t in launch()])
            finally:
                _lib.set_tuning(**old)
        return outs

    exact = run(True)
    for mode in (1, 4):
        for a, b in zip(exact[0][0], exact[mode][0]):
            assert torch.equal(a, b), (case, mode)
    rnd = run(False)
    for a, b in zip(rnd[1][0], rnd[1][1]):
        assert torch.equal(a, b), case                    # repeatable
    for a, b in zip(rnd[0][0], rnd[1][0]):
        assert _rel_err(a.float(), b.float()) < 1.6e-2, case


@pytest.mark.parametrize("M,N,K,code", [(70000, 256, 576, 256257), (66000, 256, 576, 272256), (33792, 512, 640, 272256),
                                        (65536 + 256, 256, 512, 272256)])
def test_gemm_nt_head_tail_split(M, N, K, code):
    """Problems a few tiles over whole rounds of 256x256 tiles on the 256 CUs either run as ONE round of 272-row tiles
    (when that covers them: 33792 x 512 = 250 tiles) or are cut into a big-tile head and a 4-wave tail
    (dispatch_nt_store): every output row (ragged last tile included) against an f64 reference, and bit-identical to
    the 4-wave kernel alone (same bf16 operands, same f32 accumulation order per element)."""
    from megreader_amd._lib import load
    lib = load()
    dtype = torch.bfloat16
    assert lib.mr_nt_kernel_code(1, M, N, K, 0) == code, "unexpected kernel choice for this shape"
    g = torch.Generator().manual_seed(M)
    A = (torch.randn(M, K, generator=g) * 0.5).to(DEV, dtype)
    B = (torch.randn(N, K, generator=g) * 0.5).to(DEV, dtype)
    bias = torch.randn(N, generator=g).to(DEV)
    C = torch.full((M, N), 7.0, device=DEV, dtype=dtype)
    call("mr_gemm_nt", 1, ptr(A), K, ptr(B), K, ptr(C), N, ptr(bias), 1, M, N, K)
    ref = torch.relu(A.double() @ B.double().t() + bias.double())
    assert _rel_err(C, ref) < _tol(dtype, K)
    old = lib.mr_set_nt_big(-1)
    try:
        C2 = torch.full((M, N), 7.0, device=DEV, dtype=dtype)
        call("mr_gemm_nt", 1, ptr(A), K, ptr(B), K, ptr(C2), N, ptr(bias), 1, M, N, K)
    finally:
        lib.mr_set_nt_big(old)
    assert torch.equal(C, C2)
    # the v3 (2-phase) and the phased (igemm_p8.h) 256x256 kernels: same operands, same per-element k order
    oldp = lib.mr_set_nt_p8(1)
    try:
        C3 = torch.full((M, N), 7.0, device=DEV, dtype=dtype)
        call("mr_gemm_nt", 1, ptr(A), K, ptr(B), K, ptr(C3), N, ptr(bias), 1, M, N, K)
    finally:
        lib.mr_set_nt_p8(oldp)
    assert torch.equal(C, C3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("P,NA,NB,perm", [(512, 128, 128, 0), (8448, 2048, 256, 256), (1000, 40, 512, 0),
                                          (77, 256, 72, 0), (4096, 64, 576, 0)])
def test_gemm_tn(dtype, P, NA, NB, perm):
    g = torch.Generator().manual_seed(P + NA)
    A = torch.randn(P, NA, generator=g)
    B = torch.randn(P, NB, generator=g)
    Ad, Bd = A.to(DEV, dtype), B.to(DEV, dtype)
    C = torch.ones(NA, NB, device=DEV)  # accumulate semantics: starts at 1
    cs = torch.full((NA,), 2.0, device=DEV)
    call("mr_gemm_tn", dtype_code(dtype), ptr(Ad), NA, ptr(Bd), NB, ptr(C), NB, P, NA, NB, perm, ptr(cs))
    ref = Ad.double().cpu().t() @ Bd.double().cpu()
    csr = Ad.double().cpu().sum(dim=0)
    if perm:
        blocks = ref.view(NA // (4 * perm), perm, 4, NB)          # row r = 4*j + q  ->  q*perm + j
        ref = blocks.permute(0, 2, 1, 3).reshape(NA, NB)
        csr = csr.view(NA // (4 * perm), perm, 4).permute(0, 2, 1).reshape(NA)
    ref = ref + 1.0
    assert _rel_err(C, ref) < (_tol(dtype, P) if dtype == torch.float32 else 3e-3)
    assert _rel_err(cs, csr + 2.0) < 1e-4


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 32, 64, 3, 64, 3, 1, 1, 1),       # CRNN conv0 (channel-padded input, no dgrad)
    (2, 16, 32, 64, 128, 3, 1, 1, 1),     # CRNN conv1
    (3, 2, 18, 512, 512, 2, 1, 0, 1),     # CRNN conv6 (2x2, pad 0)
    (2, 9, 13, 32, 48, 3, 2, 1, 1),       # strided (ResNet)
    (2, 12, 10, 16, 24, 3, 1, 2, 2),      # dilated (ResnetDilated)
    (2, 8, 8, 64, 256, 1, 1, 0, 1),       # 1x1
    (1, 15, 17, 24, 40, (2, 3), (2, 1), (0, 1), 1),  # attention encoder's (2,3) kernel, stride (2,1), pad (0,1)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(dtype, case):
    N, H, W, Cin, Cout, k, s, p, d = case
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    conv = torch.nn.Conv2d(Cin, Cout, k, s, p, d)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
        conv.bias.copy_(torch.randn(Cout, generator=g))
    need_dx = Cin % 8 == 0
    # reference on (bf16-rounded) operands, f64 math
    xr = x.to(dtype).double().requires_grad_(need_dx)
    wr = conv.weight.detach().to(dtype).double().requires_grad_(True)
    br = conv.bias.detach().double().requires_grad_(True)
    yr = torch.relu(TF.conv2d(xr, wr, br, conv.stride, conv.padding, conv.dilation))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.to(dtype).double())

    xd = x.to(DEV)
    if need_dx:
        xd = xd.to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = conv.weight.detach().to(DEV).requires_grad_(True)
    b = conv.bias.detach().to(DEV).requires_grad_(True)
    y = F.conv2d(xd, w, b, conv.stride, conv.padding, conv.dilation, relu=True)
    assert y.shape == yr.shape and y.dtype == dtype
    tol = _tol(dtype, Cin * 9)
    assert _rel_err(y, yr) < tol
    y.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    assert _rel_err(w.grad, wr.grad) < (tol if dtype == torch.float32 else 2e-2)
    assert _rel_err(b.grad, br.grad) < (tol if dtype == torch.float32 else 2e-2)
    if need_dx:
        assert _rel_err(xd.grad, xr.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm(dtype, relu):
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 4, 64, 6, 9
    x = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(dtype)
    bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g))
    xr = x.double().requires_grad_(True)
    yr = bn(xr)
    if relu:
        yr = torch.relu(yr)
    gy = torch.randn(yr.shape, generator=g).to(dtype)
    yr.backward(gy.double())

    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gamma = bn.weight.detach().float().to(DEV).requires_grad_(True)
    beta = bn.bias.detach().float().to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y = F.batch_norm(xd, gamma, beta, rm, rv, True, 0.1, 1e-5, relu=relu)
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2
    assert _rel_err(y, yr) < tol
    assert _rel_err(rm, bn.running_mean) < 1e-5 and _rel_err(rv, bn.running_var) < 1e-5
    y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    assert _rel_err(xd.grad, xr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert _rel_err(gamma.grad, bn.weight.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    assert _rel_err(beta.grad, bn.bias.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("relu_input", [False, True])
@pytest.mark.parametrize("k,s,p", [((2, 2), (2, 2), (0, 0)), ((2, 2), (2, 1), (0, 1)), ((3, 3), (2, 2), (1, 1)),
                                   ((3, 2), (1, 2), (1, 0))])
def test_maxpool(dtype, k, s, p, relu_input):
    """Forward bit-exact (first-max rule on ties); backward in the fixed-geometry kernels (2x2/2, 2x2/(2,1), 3x3/2) and the
    generic one (3x2/(1,2)); relu_input=True also applies the mask of the ReLU that produced the pool input."""
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 16, 10, 14, generator=g).to(dtype)
    x = torch.relu(x)  # plenty of exact ties at 0: exercises the first-max rule
    xr = x.double().requires_grad_(True)
    yr = TF.max_pool2d(xr, k, s, p)
    gy = torch.randn(yr.shape, generator=g).to(dtype)
    yr.backward(gy.double())
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.max_pool2d(xd, k, s, p, relu_input=relu_input)
    assert torch.equal(y.float().cpu(), yr.float())
    y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    want = xr.grad * (x.double() > 0) if relu_input else xr.grad
    assert _rel_err(xd.grad, want) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,s,p,shape", [((2, 2), (2, 2), (0, 0), (5, 128, 16, 64)), ((2, 2), (2, 2), (0, 0), (3, 64, 10, 14)),
                                         ((2, 2), (2, 2), (0, 0), (2, 64, 9, 13)),      # odd sizes: uncovered last row / column
                                         ((2, 2), (2, 1), (0, 1), (4, 256, 8, 32)), ((2, 2), (2, 1), (0, 1), (3, 64, 4, 33)),
                                         ((2, 2), (2, 1), (0, 0), (2, 64, 6, 9)), ((3, 3), (2, 2), (1, 1), (2, 64, 17, 31))])
def test_maxpool_round5_kernels_are_bit_identical(dtype, k, s, p, shape):
    """mr_tuning.pool_fixed: the fixed-geometry forward (packed code store) and the pooled-element-organised 2x2 / stride 2
    backward give the same bits -- values, arg-max codes (through the gradient) and ReLU masking -- as the round-4 kernels."""
    from megreader_amd import _lib
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(*shape, generator=g)).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
    out = {}
    for mode in (0, 1):
        old = _lib.set_tuning(pool_fixed=mode)
        try:
            res = []
            for relu_input in (False, True):
                xd = x.clone().requires_grad_(True)
                y = F.max_pool2d(xd, k, s, p, relu_input=relu_input)
                gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).to(dtype).to(DEV)
                y.backward(gy.contiguous(memory_format=torch.channels_last))
                res += [y.detach().float().cpu(), xd.grad.float().cpu()]
            out[mode] = res
        finally:
            _lib.set_tuning(**old)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cout,sinks", [(64, False), (64, True), (1, False), (1, True), (24, False)])
def test_conv_transpose2x2(dtype, cout, sinks):
    """nn.ConvTranspose2d(cin, cout, 2, 2) of the DB heads (reference decoders/seg_detector.py:66-79) as GEMM + depth-to-space in
    ONE autograd node, against torch's conv_transpose2d in float64; with the fused optimizer's gradient sinks the weight / bias
    gradients land in the flat buffer (deferred, grouped weight-gradient launch included)."""
    from megreader_amd.decoders.seg_detector import ConvTranspose2x2
    from megreader_amd.optim import FusedSGD
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(17 + cout)
    N, cin, H, W = 2, 64, 5, 7
    x = torch.randn(N, cin, H, W, generator=g).to(dtype)
    mod = ConvTranspose2x2(cin, cout)
    w = mod.weight.detach().clone()
    b = mod.bias.detach().clone()
    xr = x.double().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = TF.conv_transpose2d(xr, wr, br, stride=2)
    gy = torch.randn(yr.shape, generator=g).to(dtype)
    yr.backward(gy.double())
    mod = mod.to(DEV)
    opt = FusedSGD(mod.parameters(), lr=0.0) if sinks else None
    if opt is not None:
        opt.zero_grad()
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = mod(xd)
    assert tuple(y.shape) == tuple(yr.shape)
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2
    assert _rel_err(y, yr) < tol
    gyd = gy.to(DEV)
    y.backward(gyd.contiguous(memory_format=torch.channels_last) if cout % 8 == 0 else gyd)
    torch.cuda.synchronize()
    gtol = 1e-4 if dtype == torch.float32 else 2e-2
    assert _rel_err(xd.grad, xr.grad) < gtol
    assert _rel_err(mod.weight.grad, wr.grad) < gtol
    assert _rel_err(mod.bias.grad, br.grad) < gtol
    if sinks:
        assert mod.weight.grad.data_ptr() == mod.weight._mr_grad_sink.data_ptr()


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear(dtype):
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(3)
    T, N, K, O = 5, 7, 64, 38
    x = torch.randn(T, N, K, generator=g).to(dtype)
    lin = torch.nn.Linear(K, O)
    xr = x.double().requires_grad_(True)
    wr = lin.weight.detach().to(dtype).double().requires_grad_(True)
    br = lin.bias.detach().double().requires_grad_(True)
    yr = TF.linear(xr, wr, br)
    gy = torch.randn(yr.shape, generator=g).to(dtype)
    yr.backward(gy.double())
    xd = x.to(DEV).requires_grad_(True)
    w = lin.weight.detach().to(DEV).requires_grad_(True)
    b = lin.bias.detach().to(DEV).requires_grad_(True)
    y = F.linear(xd, w, b)
    assert y.shape == (T, N, O)
    tol = _tol(dtype, K)
    assert _rel_err(y, yr) < tol
    y.backward(gy.to(DEV))
    assert _rel_err(xd.grad, xr.grad) < tol
    assert _rel_err(w.grad, wr.grad) < (tol if dtype == torch.float32 else 2e-2)
    assert _rel_err(b.grad, br.grad) < (tol if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,N,I,H", [(5, 3, 64, 32), (9, 70, 32, 64)])
def test_bilstm(dtype, T, N, I, H):
    mr.set_compute_dtype(dtype)
    torch.manual_seed(17)
    ref = torch.nn.LSTM(I, H, bidirectional=True).double()
    x = torch.randn(T, N, I).to(dtype)
    if dtype == torch.bfloat16:  # reference sees the same bf16-rounded weights
        with torch.no_grad():
            for n_, p in ref.named_parameters():
                if n_.startswith("weight"):
                    p.copy_(p.float().to(dtype).double())
    xr = x.double().requires_grad_(True)
    yr, _ = ref(xr)
    gy = torch.randn(yr.shape).to(dtype)
    yr.backward(gy.double())
    params = [getattr(ref, n_).detach().float().to(DEV).requires_grad_(True) for n_ in
              ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
               "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse")]
    xd = x.to(DEV).requires_grad_(True)
    y = F.bilstm(xd, *params)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    assert _rel_err(y, yr) < tol
    y.backward(gy.to(DEV))
    assert _rel_err(xd.grad, xr.grad) < (2e-4 if dtype == torch.float32 else 4e-2)
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
             "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse")
    for n_, p in zip(names, params):
        assert _rel_err(p.grad, getattr(ref, n_).grad) < (2e-4 if dtype == torch.float32 else 4e-2), n_


@pytest.mark.parametrize("T,N,I", [(33, 256, 512), (33, 256, 256), (7, 40, 64), (2, 16, 64), (1, 5, 64)])
def test_bilstm_persistent_recurrence(T, N, I):
    """The one-launch persistent recurrence (csrc/lstm_persist.hip: W_hh slices in registers, h all-gathered /
    dh reduce-scattered between workgroups through tagged granules) against (a) the per-step launches it replaces,
    same bf16 operands and f32 accumulation, every output and every gradient element-wise, and (b) a float64 torch
    LSTM.  Shapes: the CRNN layers at the benchmarked batch (both input widths), a ragged batch (40 = 2.5 groups of
    16 rows), and the T=2 / T=1 edge cases (one / no exchange)."""
    from megreader_amd._lib import load
    from megreader_amd.nn import functional as Fn
    H = 256
    dtype = torch.bfloat16
    mr.set_compute_dtype(dtype)
    lib = load()
    assert lib.mr_lstm_ws_bytes(1, T, N, H) > 0, "persistent path must apply at H=256 in bf16"
    assert lib.mr_lstm_ws_bytes(0, T, N, H) == 0, "f32 parity mode runs the per-step kernels"
    torch.manual_seed(3)
    ref = torch.nn.LSTM(I, H, bidirectional=True).double()
    with torch.no_grad():
        for n_, p in ref.named_parameters():
            if n_.startswith("weight"):
                p.copy_(p.float().to(dtype).double())
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
             "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse")
    x = torch.randn(T, N, I).to(dtype)
    gy = torch.randn(T, N, 2 * H).to(dtype)

    def run(persist):
        lib.mr_set_lstm_persist(persist)
        params = [getattr(ref, n_).detach().float().to(DEV).requires_grad_(True) for n_ in names]
        xd = x.to(DEV).requires_grad_(True)
        y = F.bilstm(xd, *params)
        y.backward(gy.to(DEV))
        torch.cuda.synchronize()
        return [y.detach().float().cpu(), xd.grad.float().cpu()] + [p.grad.float().cpu() for p in params]

    Fn.LSTM_STATUS = []
    try:
        got = run(1)
        assert len(Fn.LSTM_STATUS) == 2, "forward and backward must both have taken the persistent path"
        assert all(int(s_.view(torch.int32).item()) == 0 for s_ in Fn.LSTM_STATUS), "a bounded spin timed out"
    finally:
        Fn.LSTM_STATUS = None
    try:
        want = run(0)
    finally:
        lib.mr_set_lstm_persist(1)
    labels = ["y", "dx"] + list(names)
    for lab, a, b in zip(labels, got, want):
        # same operands, same accumulation precision; the v_exp/v_rcp gate functions and the summation order of the
        # recurrent GEMM differ in the last f32 bits, which bf16 rounding of h occasionally turns into one bf16 ulp
        assert _rel_err(a, b) < 6e-3, (lab, _rel_err(a, b))
    xr = x.double().requires_grad_(True)
    yr, _ = ref(xr)
    yr.backward(gy.double())
    assert _rel_err(got[0], yr) < 2e-2
    assert _rel_err(got[1], xr.grad) < 4e-2
    for n_, a in zip(names, got[2:]):
        assert _rel_err(a, getattr(ref, n_).grad) < 4e-2, n_


@pytest.mark.parametrize("dtype", DTYPES)
def test_ctc_matches_torch_and_oracle(dtype):
    from oracle.ctc import ctc_1d
    g = torch.Generator().manual_seed(23)
    T, N, C, S = 12, 6, 38, 32
    logits = (torch.randn(T, N, C, generator=g) * 2).to(dtype)
    lengths = torch.tensor([3, 1, 6, 0, 10, 5])
    targets = torch.zeros(N, S, dtype=torch.int32)
    for i, L in enumerate(lengths.tolist()):
        targets[i, :L] = torch.randint(2, C, (L,), generator=g, dtype=torch.int32)
    targets[2, 1] = targets[2, 0]            # repeated label
    targets[4, :10] = 5                       # 10 repeats need T >= 19 > 12: infeasible -> zero_infinity path
    xr = logits.float().clone().requires_grad_(True)
    lp = TF.log_softmax(xr, dim=2).double()
    loss_r = TF.ctc_loss(lp, targets, torch.full((N,), T, dtype=torch.int32), lengths, zero_infinity=True)
    loss_r.backward()
    xd = logits.to(DEV).requires_grad_(True)
    loss, logp = F.ctc_loss_logits(xd, targets.to(DEV), None, lengths.to(DEV))
    assert loss.dtype == torch.float64
    assert abs(float(loss) - float(loss_r)) < 1e-6 * max(1.0, abs(float(loss_r)))
    assert float((logp.cpu() - lp.float()).abs().max()) < 2e-6
    loss.backward()
    gtol = 2e-6 if dtype == torch.float32 else 4e-3
    assert float((xd.grad.float().cpu() - xr.grad).abs().max()) < gtol
    o = ctc_1d(logits.float().numpy(), targets.numpy(), lengths.numpy())
    assert abs(o['loss'] - float(loss)) < 1e-6 * max(1.0, abs(o['loss']))  # f32 log-softmax round-off
    assert float((torch.from_numpy(o['grad_logits']).float() - xd.grad.float().cpu()).abs().max()) < gtol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scale", [1.0, 8.0])
def test_ctc_scaled_linear_domain_equals_log_domain(dtype, scale):
    """mr_tuning.ctc_linear: the scaled linear-domain recursions (round 5) against the float64 log-sum-exp kernels they
    replace -- ragged input lengths, an empty target, repeated labels, an infeasible target (zero_infinity), peaked logits."""
    from megreader_amd import _lib
    g = torch.Generator().manual_seed(29)
    T, N, C, S = 33, 9, 38, 25
    logits = (torch.randn(T, N, C, generator=g) * scale).to(dtype)
    lengths = torch.tensor([3, 1, 6, 0, 10, 5, 25, 12, 2])
    in_len = torch.tensor([33, 20, 33, 5, 12, 33, 33, 30, 1])
    targets = torch.zeros(N, S, dtype=torch.int64)
    for i, L in enumerate(lengths.tolist()):
        targets[i, :L] = torch.randint(1, C, (L,), generator=g)
    targets[2, 1] = targets[2, 0]
    targets[4, :10] = 5                        # 10 repeats in 12 frames: infeasible
    out = {}
    for mode in (0, 1):
        old = _lib.set_tuning(ctc_linear=mode)
        try:
            xd = logits.to(DEV).requires_grad_(True)
            loss, logp = F.ctc_loss_logits(xd, targets.to(DEV), in_len.to(DEV), lengths.to(DEV))
            loss.backward()
            per, _ = F.ctc_loss_logits(xd.detach(), targets.to(DEV), in_len.to(DEV), lengths.to(DEV), per_sample=True,
                                       zero_infinity=False)
            out[mode] = (float(loss), xd.grad.float().cpu().clone(), logp.cpu().clone(), per.cpu().clone())
        finally:
            _lib.set_tuning(**old)
    # (the two kernels sum the exponentials of the f32 log-softmax in different orders: the log-probabilities may differ in
    # the last place, 33 of them add up in a loss; everything downstream is float64 in both)
    assert float((out[0][2] - out[1][2]).abs().max()) <= 4e-6
    assert abs(out[0][0] - out[1][0]) <= 2e-6 * max(1.0, abs(out[0][0]))
    assert torch.equal(torch.isinf(out[0][3]), torch.isinf(out[1][3]))
    fin = ~torch.isinf(out[0][3])
    assert float((out[0][3][fin] - out[1][3][fin]).abs().max()) < 1e-4
    assert float((out[0][1] - out[1][1]).abs().max()) <= (2e-6 if dtype == torch.float32 else 1e-4)


def test_ctc_full_size_properties():
    """BASELINE size (T=33, N=256, C=38): occupancy rows of the gradient sum to zero, loss finite and positive."""
    g = torch.Generator().manual_seed(1)
    T, N, C, S = 33, 256, 38, 32
    logits = torch.randn(T, N, C, generator=g).to(DEV).requires_grad_(True)
    lengths = torch.randint(3, 11, (N,), generator=g)
    targets = torch.zeros(N, S, dtype=torch.int32)
    for i, L in enumerate(lengths.tolist()):
        targets[i, :L] = torch.randint(2, C, (L,), generator=g, dtype=torch.int32)
    loss, _ = F.ctc_loss_logits(logits, targets.to(DEV), None, lengths.to(DEV))
    loss.backward()
    assert math.isfinite(float(loss)) and float(loss) > 0
    assert float(logits.grad.sum(dim=2).abs().max()) < 1e-6   # sum_c (softmax - occupancy) = 0 for every (t, n)


def test_cpu_tensor_fails_loudly():
    with pytest.raises(NotImplementedError):
        F.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 3, 3))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_relu_pool_fused_backward(dtype):
    """conv(+ReLU, mask deferred) -> max-pool(relu_input): the pool's backward applies the ReLU mask."""
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 16, 8, 10, generator=g).to(dtype)
    w = (torch.randn(24, 16, 3, 3, generator=g) * 0.2)
    b = torch.randn(24, generator=g)
    xr = x.double().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = TF.max_pool2d(torch.relu(TF.conv2d(xr, wr, br, 1, 1)), (2, 2), (2, 1), (0, 1))
    gy = torch.randn(yr.shape, generator=g).to(dtype)
    yr.backward(gy.double())
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    y = F.max_pool2d(F.conv2d(xd, wd, bd, (1, 1), (1, 1), (1, 1), relu=True, relu_grad_downstream=True),
                     (2, 2), (2, 1), (0, 1), relu_input=True)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert _rel_err(y, yr) < tol
    y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    assert _rel_err(xd.grad, xr.grad) < tol
    assert _rel_err(wd.grad, wr.grad) < tol
    assert _rel_err(bd.grad, br.grad) < tol


@pytest.mark.parametrize("mode", [1, 2, 3, 6, 7])
def test_big_tile_nt_kernel_bit_identical_to_4wave_kernel(mode):
    """The 8-wave 256x256 / 288x256 / 272x256 NT kernel (picked automatically for CU-filling shapes) against the 4-wave
    kernel on the same operands: same MFMA instruction, same k order per output element -> bit-identical results.
    Covers dense GEMM, conv forward and conv dgrad operands, ragged last row tiles and two column tiles."""
    from megreader_amd._lib import load
    lib = load()
    dtype = torch.bfloat16
    dt = dtype_code(dtype)
    g = torch.Generator().manual_seed(5 + mode)

    def both(fn):
        from megreader_amd import _lib as _l
        old = lib.mr_set_nt_big(-1)
        # (same summation order on both sides: the 4-wave kernel's split reduction, mr_tuning.nt_ksplit, would cut the k-loop of
        # these few-tile problems)
        old_split = _l.set_tuning(nt_ksplit=0)
        try:
            ref = fn()
            lib.mr_set_nt_big(mode)
            out = fn()
        finally:
            lib.mr_set_nt_big(old)
            _l.set_tuning(**old_split)
        return ref, out

    # dense: M ragged against 256 and 288, N = 512 (two column tiles), K not a multiple of 64
    M, N, K = 1000, 512, 1096
    A = torch.randn(M, K, generator=g).to(DEV, dtype)
    B = torch.randn(N, K, generator=g).to(DEV, dtype)
    bias = torch.randn(N, generator=g).to(DEV)

    def dense():
        C = torch.zeros(M, N, device=DEV, dtype=dtype)
        call("mr_gemm_nt", dt, ptr(A), K, ptr(B), K, ptr(C), N, ptr(bias), 1, M, N, K)
        return C

    ref, out = both(dense)
    assert torch.equal(ref, out)
    want = torch.relu(A.double().cpu() @ B.double().cpu().t() + bias.double().cpu())
    assert _rel_err(out, want) < _tol(dtype, K)

    # conv forward / dgrad (3x3, pad 1 and the 2x2 pad 0 head), C = K = 256
    for (Nb, H, W, C, Kc, k, p) in [(5, 8, 32, 256, 256, 3, 1), (9, 2, 34, 256, 512, 2, 0)]:
        Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
        x = torch.randn(Nb, H, W, C, generator=g).to(DEV, dtype)
        w = (torch.randn(Kc, k, k, C, generator=g) * 0.05).to(DEV, dtype)
        wt = (torch.randn(C, k, k, Kc, generator=g) * 0.05).to(DEV, dtype)
        dy = torch.randn(Nb, Ho, Wo, Kc, generator=g).to(DEV, dtype)
        cb = torch.randn(Kc, generator=g).to(DEV)

        def fwd():
            y = torch.zeros(Nb, Ho, Wo, Kc, device=DEV, dtype=dtype)
            call("mr_conv2d_fwd", dt, ptr(x), ptr(w), ptr(cb), ptr(y), 1, Nb, H, W, C, C, Kc, Kc, k, k, 1, 1, p, p, 1, 1,
                 Ho, Wo)
            return y

        def dgrad():
            dx = torch.zeros(Nb, H, W, C, device=DEV, dtype=dtype)
            call("mr_conv2d_dgrad", dt, ptr(dy), ptr(wt), ptr(dx), Nb, H, W, C, C, Kc, Kc, k, k, 1, 1, p, p, 1, 1, Ho,
                 Wo)
            return dx

        for fn in (fwd, dgrad):
            ref, out = both(fn)
            assert torch.equal(ref, out), (fn.__name__, Nb, H, W, C, Kc, k)
        want = torch.relu(TF.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2),
                                    cb.double().cpu(), 1, p)).permute(0, 2, 3, 1)
        assert _rel_err(fwd(), want) < _tol(dtype, C * k * k)


@pytest.mark.parametrize("mode", [1, 2])
def test_big_tile_tn_kernel_matches_128_tile_kernel(mode):
    """The wide-tile TN (weight-gradient) kernels (mode 1: 256x256, mode 2: 128x256) against the 128x128 kernel and
    an f64 reference: dense with
    gate-interleaved row permutation + fused column sums, conv wgrad (3x3 pad 1; 2x2 pad 0) with the fused bias
    gradient, ragged NB (not a multiple of 256) and P not a multiple of the 64-row step."""
    from megreader_amd._lib import load
    lib = load()
    dtype = torch.bfloat16
    dt = dtype_code(dtype)
    g = torch.Generator().manual_seed(11)

    def both(fn):
        old = lib.mr_set_tn_big(-1)
        try:
            ref = fn()
            lib.mr_set_tn_big(mode)
            out = fn()
        finally:
            lib.mr_set_tn_big(old)
        return ref, out

    # dense: NA = 512 with the LSTM row permutation (perm = 64), NB = 320 (ragged), P = 1000
    P, NA, NB, perm = 1000, 512, 320, 64
    A = torch.randn(P, NA, generator=g).to(DEV, dtype)
    B = torch.randn(P, NB, generator=g).to(DEV, dtype)

    def dense():
        C = torch.ones(NA, NB, device=DEV)
        cs = torch.full((NA,), 2.0, device=DEV)
        call("mr_gemm_tn", dt, ptr(A), NA, ptr(B), NB, ptr(C), NB, P, NA, NB, perm, ptr(cs))
        return C, cs

    (c0, s0), (c1, s1) = both(dense)
    ref = A.double().cpu().t() @ B.double().cpu()
    ref = ref.view(NA // (4 * perm), perm, 4, NB).permute(0, 2, 1, 3).reshape(NA, NB) + 1.0
    csr = A.double().cpu().sum(dim=0).view(NA // (4 * perm), perm, 4).permute(0, 2, 1).reshape(NA) + 2.0
    assert _rel_err(c1, ref) < 2e-5 and _rel_err(c0, ref) < 2e-5
    assert _rel_err(s1, csr) < 2e-5 and _rel_err(s0, csr) < 2e-5

    for (Nb, H, W, C, Kc, k, p) in [(3, 8, 32, 256, 256, 3, 1), (7, 2, 34, 256, 512, 2, 0)]:
        Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
        x = torch.randn(Nb, H, W, C, generator=g).to(DEV, dtype)
        dy = torch.randn(Nb, Ho, Wo, Kc, generator=g).to(DEV, dtype)

        def wgrad():
            gw = torch.zeros(Kc, k, k, C, device=DEV)
            gb = torch.zeros(Kc, device=DEV)
            call("mr_conv2d_wgrad", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), Nb, H, W, C, C, Kc, Kc, k, k, 1, 1, p, p, 1, 1,
                 Ho, Wo)
            return gw, gb

        (w0, b0), (w1, b1) = both(wgrad)
        tab = torch.empty(Nb * Ho * Wo, 2, dtype=torch.int32, device=DEV)

        def wgrad_tab():   # the row-table path (what the training step uses) through the wide-tile kernel
            gw = torch.zeros(Kc, k, k, C, device=DEV)
            gb = torch.zeros(Kc, device=DEV)
            call("mr_conv2d_wgrad_tab", dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), Nb, H, W, C, C, Kc, Kc, k, k, 1, 1, p, p,
                 1, 1, Ho, Wo, ptr(tab), 1)
            return gw, gb

        (_, _), (w2, b2) = both(wgrad_tab)
        xr = x.double().cpu().permute(0, 3, 1, 2)
        wr = torch.zeros(Kc, C, k, k, dtype=torch.float64, requires_grad=True)
        TF.conv2d(xr, wr, None, 1, p).backward(dy.double().cpu().permute(0, 3, 1, 2))
        want = wr.grad.permute(0, 2, 3, 1)
        assert _rel_err(w1, want) < 2e-5 and _rel_err(w0, want) < 2e-5, (Nb, H, W)
        bsum = dy.double().cpu().sum(dim=(0, 1, 2))
        assert _rel_err(b1, bsum) < 2e-5 and _rel_err(b0, bsum) < 2e-5
        assert _rel_err(w2, want) < 2e-5 and _rel_err(b2, bsum) < 2e-5, ("row table", Nb, H, W)


def test_conv_wgrad_row_table_matches_plain_wgrad():
    """mr_conv2d_wgrad_tab (per-pixel row table of the gather, built once) against mr_conv2d_wgrad: strided, dilated,
    asymmetric padding, 2x2 without padding, P not a multiple of the 64-row step; second call re-uses the table."""
    dtype = torch.bfloat16
    dt = dtype_code(dtype)
    g = torch.Generator().manual_seed(23)
    cases = [(3, 8, 32, 64, 72, 3, 3, 1, 1, 1, 1, 1, 1), (2, 9, 11, 16, 40, 3, 3, 2, 2, 1, 1, 1, 1),
             (5, 2, 34, 128, 128, 2, 2, 1, 1, 0, 0, 1, 1), (2, 12, 10, 32, 24, 3, 3, 1, 1, 2, 2, 2, 2),
             (2, 7, 9, 8, 16, 3, 1, 1, 2, 1, 0, 1, 1)]
    for (Nb, H, W, C, Kc, R, S, sh, sw, ph, pw, dh, dw) in cases:
        Ho = (H + 2 * ph - dh * (R - 1) - 1) // sh + 1
        Wo = (W + 2 * pw - dw * (S - 1) - 1) // sw + 1
        x = torch.randn(Nb, H, W, C, generator=g).to(DEV, dtype)
        dy = torch.randn(Nb, Ho, Wo, Kc, generator=g).to(DEV, dtype)
        tab = torch.empty(Nb * Ho * Wo, 2, dtype=torch.int32, device=DEV)

        def run(name, *extra):
            gw = torch.zeros(Kc, R, S, C, device=DEV)
            gb = torch.zeros(Kc, device=DEV)
            call(name, dt, ptr(dy), ptr(x), ptr(gw), ptr(gb), Nb, H, W, C, C, Kc, Kc, R, S, sh, sw, ph, pw, dh, dw, Ho,
                 Wo, *extra)
            return gw, gb

        w0, b0 = run("mr_conv2d_wgrad")
        w1, b1 = run("mr_conv2d_wgrad_tab", ptr(tab), 1)
        w2, b2 = run("mr_conv2d_wgrad_tab", ptr(tab), 0)
        case = (Nb, H, W, C, Kc, R, S, sh, sw, ph, pw, dh, dw)
        assert _rel_err(w1, w0) < 2e-5 and _rel_err(w2, w0) < 2e-5, case
        assert _rel_err(b1, b0) < 2e-5 and _rel_err(b2, b0) < 2e-5, case
        xr = x.double().cpu().permute(0, 3, 1, 2)
        wr = torch.zeros(Kc, C, R, S, dtype=torch.float64, requires_grad=True)
        TF.conv2d(xr, wr, None, (sh, sw), (ph, pw), (dh, dw)).backward(dy.double().cpu().permute(0, 3, 1, 2))
        assert _rel_err(w1, wr.grad.permute(0, 2, 3, 1)) < 2e-5, case


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tn_row_count_not_a_vector_multiple(dtype):
    """mr_gemm_tn with NA = 38 rows of A^T B taken from a 40-column A (a Linear layer with 38 outputs accumulating
    straight into its [38, K] gradient sink): rows 38, 39 of the padded product are never stored, the fused column sums
    stop at 38 as well."""
    dt = dtype_code(dtype)
    g = torch.Generator().manual_seed(5)
    P, NA, lda, NB = 777, 38, 40, 64
    A = torch.randn(P, lda, generator=g).to(DEV, dtype)
    B = torch.randn(P, NB, generator=g).to(DEV, dtype)
    C = torch.full((NA + 2, NB), 3.0, device=DEV)      # two guard rows behind the sink
    cs = torch.full((NA + 2,), 5.0, device=DEV)
    call("mr_gemm_tn", dt, ptr(A), lda, ptr(B), NB, ptr(C), NB, P, NA, NB, 0, ptr(cs))
    ref = A[:, :NA].double().cpu().t() @ B.double().cpu() + 3.0
    assert _rel_err(C[:NA], ref) < (2e-5 if dtype == torch.float32 else 2e-5)
    assert torch.equal(C[NA:].cpu(), torch.full((2, NB), 3.0)) and torch.equal(cs[NA:].cpu(), torch.full((2,), 5.0))
    assert _rel_err(cs[:NA], A[:, :NA].double().cpu().sum(0) + 5.0) < 2e-5


@pytest.mark.parametrize("group", [-2, 0, 1, 2, 3, 4, 16])
@pytest.mark.parametrize("P,NA,NB,splits", [(8448, 256, 512, 4), (8448, 256, 512, 7), (4096, 40, 136, 3), (33000, 128, 128, 16)])
def test_gemm_tn_split_group_reduction_exact(group, P, NA, NB, splits):
    """In-launch reduction of the TN GEMM kernel's split partials (TnArgs.grp: sc1 slabs + ticket, the last arriver sums; plain
    read-modify-write when one group holds all the splits, atomics otherwise).  Integer operands: every partial sum is
    exact, so C (accumulate semantics, starts at 1) and the column sums must EQUAL the float64 result for any grouping,
    and a second launch (tickets reset by the first) must add the same amount again.
    group == -2: mr_set_tn_fin(2) -- every workgroup's partial tile goes to its slab with plain stores and a finalize launch
    adds the sum over the splits into C (opt-in); the other values run the in-launch group reduction (fin 0, the default)."""
    from megreader_amd import _lib
    lib = _lib.load()
    _lib.ensure_tn_workspace(DEV)
    oldf = lib.mr_set_tn_fin(2 if group == -2 else 0)
    group = max(group, 0)
    g = torch.Generator().manual_seed(P + NA + group)
    A = torch.randint(-3, 4, (P, NA), generator=g).float()
    B = torch.randint(-3, 4, (P, NB), generator=g).float()
    lda = (NA + 7) // 8 * 8
    Ad = torch.zeros(P, lda).copy_(torch.nn.functional.pad(A, (0, lda - NA))).to(DEV, torch.bfloat16)
    Bd = B.to(DEV, torch.bfloat16)
    C = torch.ones(NA, NB, device=DEV)
    cs = torch.zeros(NA, device=DEV)
    oldg, olds = lib.mr_set_tn_group(group), lib.mr_set_tn_splits(splits)
    try:
        for _ in range(2):
            call("mr_gemm_tn", 1, ptr(Ad), lda, ptr(Bd), NB, ptr(C), NB, P, NA, NB, 0, ptr(cs))
        torch.cuda.synchronize()
    finally:
        lib.mr_set_tn_group(oldg)
        lib.mr_set_tn_splits(olds)
        lib.mr_set_tn_fin(oldf)
    ref = 1 + 2 * (A.double().t() @ B.double())
    assert torch.equal(C.cpu().double(), ref), float((C.cpu().double() - ref).abs().max())
    assert torch.equal(cs.cpu().double(), 2 * A.double().sum(0))


def test_conv_wgrad_gemm_kernel_group_reduction_exact():
    """Same through the conv wgrad entry for geometries the all-taps kernel does not take (2x2 head, strided 3x3, W = 64)."""
    from megreader_amd import _lib
    lib = _lib.load()
    _lib.ensure_tn_workspace(DEV)
    for (N, H, W, C, K, k, st, p) in [(16, 2, 34, 64, 64, 2, 1, 0), (8, 8, 32, 64, 128, 3, 2, 1), (4, 16, 64, 64, 128, 3, 1, 1)]:
        Ho, Wo = (H + 2 * p - k) // st + 1, (W + 2 * p - k) // st + 1
        g = torch.Generator().manual_seed(N + H)
        x = torch.randint(-3, 4, (N, H, W, C), generator=g).float()
        dy = torch.randint(-3, 4, (N, Ho, Wo, K), generator=g).float()
        wref = torch.zeros(K, C, k, k, dtype=torch.float64, requires_grad=True)
        TF.conv2d(x.permute(0, 3, 1, 2).double(), wref, None, st, p).backward(dy.permute(0, 3, 1, 2).double())
        ref = wref.grad.permute(0, 2, 3, 1)
        assert lib.mr_tn_taps_would_run(N, H, W, C, C, K, K, k, k, st, st, p, p, 1, 1, Ho, Wo) == 0
        xd, dyd = x.to(DEV, torch.bfloat16), dy.to(DEV, torch.bfloat16)
        for group in (-2, 0, 1, 2):      # -2: slabs + finalize launch (mr_set_tn_fin(2), opt-in); else in-launch groups
            oldf = lib.mr_set_tn_fin(2 if group == -2 else 0)
            old = lib.mr_set_tn_group(max(group, 0))
            try:
                gw = torch.zeros(K, k, k, C, device=DEV)
                gb = torch.zeros(K, device=DEV)
                tab = torch.empty(N * Ho * Wo, 2, dtype=torch.int32, device=DEV)
                call("mr_conv2d_wgrad_tab", 1, ptr(dyd), ptr(xd), ptr(gw), ptr(gb),
                     N, H, W, C, C, K, K, k, k, st, st, p, p, 1, 1, Ho, Wo, ptr(tab), 1)
                torch.cuda.synchronize()
            finally:
                lib.mr_set_tn_group(old)
                lib.mr_set_tn_fin(oldf)
            assert torch.equal(gw.cpu().double(), ref), (group, float((gw.cpu().double() - ref).abs().max()))
            assert torch.equal(gb.cpu().double(), dy.double().sum((0, 1, 2)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C,H,W,relu,res", [(4, 64, 6, 9, True, False), (16, 256, 4, 33, True, False), (3, 512, 5, 7, False, True),
                                              (8, 128, 8, 32, True, True), (2, 2048, 2, 3, False, False)])
def test_batchnorm_fused_finalize_bit_identical(dtype, N, C, H, W, relu, res):
    """Training-mode BN with the finalize kernels folded into the apply passes (mr_set_bn_fused(1), the default) against
    the separate-launch path on the same inputs: outputs, saved / running statistics, step counter, dx, the residual
    gradient, dgamma and dbeta must be bit-identical (same double-precision per-channel expressions, same f32 maps)."""
    from megreader_amd._lib import load
    lib = load()
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(C + H)
    x0 = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
    r0 = torch.randn(N, C, H, W, generator=g).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last) if res else None
    gy = torch.randn(N, C, H, W, generator=g).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
    w0, b0 = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    outs = []
    from megreader_amd._lib import set_tuning
    for mode in (1, 0):
        old = lib.mr_set_bn_fused(mode)
        old1 = set_tuning(bn_onepass=0)      # the one-pass backward has its own test below (other summation order)
        try:
            xd = x0.clone().requires_grad_(True)
            rd = r0.clone().requires_grad_(True) if res else None
            gamma, beta = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            nbt = torch.zeros((), dtype=torch.int64, device=DEV)
            y = F.batch_norm(xd, gamma, beta, rm, rv, True, 0.1, 1e-5, relu=relu, residual=rd, num_batches_tracked=nbt)
            y.backward(gy)
            torch.cuda.synchronize()
            outs.append([y.detach(), rm, rv, nbt, xd.grad, gamma.grad, beta.grad] + ([rd.grad] if res else []))
        finally:
            lib.mr_set_bn_fused(old)
            set_tuning(**old1)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert int(outs[0][3]) == 1


# ---------------------------------------------------------------------------------------------------------------
# One-pass BatchNorm backward (bn_bwd_onepass_kernel, mr_tuning.bn_onepass): reductions, a barrier among the workgroups of a
# 64-channel slab and dx from registers in ONE launch, against the reduction launch + apply launch on the same inputs (the
# partial sums are grouped differently, so equality is to rounding, not bit for bit) and against f64 torch.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C,H,W,relu,res", [
    (4, 64, 6, 9, True, False),          # 216 rows: one ragged workgroup (2-row variant)
    (16, 256, 4, 33, True, True),        # 2112 rows x 4 slabs, fused ReLU + residual gradient
    (3, 512, 5, 7, False, False),        # 105 rows: less than one batch of rows
    (2, 2048, 2, 3, False, True),        # 12 rows x 32 slabs
    (32, 256, 16, 64, True, True),       # 32768 x 256 = 8.4 M elements: the 8-row variant, 512 workgroups (FPN layer1 at N = 32)
    (2, 64, 320, 320, True, False),      # 204800 x 64 (DB stem): 800 workgroups of one slab (above the resident grid in bf16:
                                         #   falls back to the two launches there, runs in one pass in f32 if it fits)
    (7, 128, 37, 41, True, True),        # ragged everywhere
])
def test_batchnorm_backward_one_pass(dtype, N, C, H, W, relu, res):
    from megreader_amd._lib import set_tuning
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(N + C + H)
    x0 = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
    r0 = torch.randn(N, C, H, W, generator=g).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last) if res else None
    gy = torch.randn(N, C, H, W, generator=g).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
    w0, b0 = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    outs = []
    for mode in (1, 0, 1):
        old = set_tuning(bn_onepass=mode)
        try:
            xd = x0.clone().requires_grad_(True)
            rd = r0.clone().requires_grad_(True) if res else None
            gamma, beta = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            y = F.batch_norm(xd, gamma, beta, rm, rv, True, 0.1, 1e-5, relu=relu, residual=rd)
            y.backward(gy)
            torch.cuda.synchronize()
            outs.append([xd.grad.float(), gamma.grad, beta.grad] + ([rd.grad.float()] if res else []))
            y_hip = y.detach()
        finally:
            set_tuning(**old)
    for o in outs:
        assert all(bool(torch.isfinite(t).all()) for t in o)      # a barrier that timed out poisons its outputs
    tol_dx = 2e-6 if dtype == torch.float32 else 8e-3              # bf16: one rounding of a dx element may flip
    for a, b, name in zip(outs[0], outs[1], ("dx", "dgamma", "dbeta", "dres")):
        bar = tol_dx if name == "dx" else (0.0 if name == "dres" else 2e-6)
        assert _rel_err(a, b) <= bar, (name, _rel_err(a, b))
    for a, b in zip(outs[0], outs[2]):                            # and the same launch again (f64 atomics: order is not fixed)
        assert _rel_err(a, b) <= tol_dx
    # against f64 torch (the statistics of the stored, compute-dtype x)
    bn = torch.nn.BatchNorm2d(C).double().to(DEV)
    with torch.no_grad():
        bn.weight.copy_(w0.double())
        bn.bias.copy_(b0.double())
    xr = x0.double().requires_grad_(True)
    rr = r0.double().requires_grad_(True) if res else None
    yr = bn(xr)
    if res:
        yr = yr + rr
    if relu:     # the kernels' mask is `stored y > 0`: the same mask here (an f64 y within rounding of zero would flip elements)
        yr = yr * (y_hip > 0).double()
    yr.backward(gy.double())
    assert _rel_err(outs[0][0], xr.grad) < (1e-4 if dtype == torch.float32 else 3e-2)
    assert _rel_err(outs[0][1], bn.weight.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert _rel_err(outs[0][2], bn.bias.grad) < (1e-4 if dtype == torch.float32 else 2e-2)


def test_batchnorm_backward_one_pass_repeated_under_load():
    """200 back-to-back one-pass launches on fresh scratch while a second stream streams through HBM: every launch must pass its
    barrier (no NaN poison) and reproduce the first result to rounding -- with and without the pre-zeroed arena (after the
    arena is exhausted the launcher zeroes sums + arrival counters itself)."""
    from megreader_amd._lib import set_tuning, get_tuning
    assert get_tuning()["bn_onepass"] == 1
    mr.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(11)
    N, C, H, W = 32, 512, 8, 32
    x0 = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV).requires_grad_(True)
    beta = torch.randn(C, generator=g).to(DEV).requires_grad_(True)
    big = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    first = None
    for it in range(200):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                big.add_(1)
        xd = x0.clone().requires_grad_(True)
        gamma.grad = beta.grad = None
        y = F.batch_norm(xd, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), True, 0.1, 1e-5, relu=True)
        y.backward(gy)
        out = (xd.grad.float(), gamma.grad.clone(), beta.grad.clone())
        assert all(bool(torch.isfinite(t).all()) for t in out), it
        if first is None:
            first = out
        else:
            assert _rel_err(out[0], first[0]) <= 8e-3 and _rel_err(out[1], first[1]) <= 2e-6 and \
                _rel_err(out[2], first[2]) <= 2e-6, it
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------------------
# BatchNorm statistics out of the conv epilogue (mr_conv2d_fwd_stats): every NT kernel family the dispatcher can pick --
# 64/96/128-row 4-wave tiles, the 256x256 and 272x256 8-wave tiles, head + tail launches -- must leave in `sums` exactly what
# a reduction over the stored y gives (the apply pass normalises the STORED values), with and without a conv bias, ragged M.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [
    (3, 7, 9, 64, 64, 3, 1, 1),        # 189 rows: one ragged 64-row tile
    (16, 16, 32, 64, 128, 1, 0, 1),    # 1x1, 8192 x 128
    (64, 8, 32, 128, 256, 3, 1, 1),    # 16384 x 256, K = 1152
    (256, 8, 32, 256, 256, 3, 1, 1),   # 65536 x 256, K = 2304: 256 x 256 big tiles (bf16)
    (256, 4, 33, 256, 512, 3, 1, 1),   # 33792 x 512: 272-row big tiles (bf16)
    (40, 16, 32, 64, 256, 1, 0, 2),    # strided 1x1 (ResNet downsample)
    (9, 20, 20, 64, 64, 3, 2, 2),      # dilated 3x3 (dilated ResNet)
])
@pytest.mark.parametrize("with_bias", [False, True])
def test_conv_fwd_stats_epilogue(dtype, shape, with_bias):
    from megreader_amd._lib import load
    Nb, H, W, C, Kc, k, p, sd = shape
    stride, dil = (sd, 1) if k == 1 else (1, sd)
    Ho = (H + 2 * p - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * p - dil * (k - 1) - 1) // stride + 1
    dt = dtype_code(dtype)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(Nb, H, W, C, generator=g) + 0.3).to(DEV, dtype)
    w = (torch.randn(Kc, k, k, C, generator=g) * 0.05).to(DEV, dtype)
    cb = (torch.randn(Kc, generator=g) * 2).to(DEV) if with_bias else None
    nsum = load().mr_bn_scratch_doubles(Kc)
    y0 = torch.empty(Nb, Ho, Wo, Kc, device=DEV, dtype=dtype)
    call("mr_conv2d_fwd", dt, ptr(x), ptr(w), ptr(cb), ptr(y0), 0, Nb, H, W, C, C, Kc, Kc, k, k, stride, stride, p, p, dil,
         dil, Ho, Wo)
    y1 = torch.empty_like(y0)
    sums = torch.zeros(nsum, dtype=torch.float64, device=DEV)
    call("mr_conv2d_fwd_stats", dt, ptr(x), ptr(w), ptr(cb), ptr(y1), ptr(sums), Nb, H, W, C, C, Kc, k, k, stride, stride,
         p, p, dil, dil, Ho, Wo)
    assert torch.equal(y0, y1)                      # the statistics epilogue does not touch the stored values
    got = sums[:16 * Kc].view(8, 2, Kc).sum(dim=0).cpu()
    yd = y1.double().view(-1, Kc).cpu()
    want = torch.stack([yd.sum(dim=0), (yd * yd).sum(dim=0)])
    P = yd.shape[0]
    # f32 partial sums of <= 17 x 16 rows, f64 across tiles: error ~ 1e-7 * sqrt(P) * rms, far below the bar
    scale = torch.stack([yd.abs().sum(dim=0), (yd * yd).sum(dim=0)]) + 1e-30
    assert float(((got - want).abs() / scale).max()) < 2e-6, float(((got - want).abs() / scale).max())
    # the reduction pass on its own (the fallback of mr_conv2d_fwd_stats, and what mr_bn_fwd_train runs otherwise)
    sums2 = torch.zeros(nsum, dtype=torch.float64, device=DEV)
    call("mr_bn_stats", dt, ptr(y1), ptr(sums2), P, Kc)
    got2 = sums2[:16 * Kc].view(8, 2, Kc).sum(dim=0).cpu()
    assert float(((got2 - want).abs() / scale).max()) < 2e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_bn_statistics_handoff_equals_separate_pass(dtype):
    """nn.Conv2d -> nn.BatchNorm2d in training mode: the first forward marks the convolution, from the second on the batch
    statistics come out of the convolution's epilogue (BatchNorm skips its reduction pass).  Outputs, running statistics
    and all gradients of a fused iteration must equal those of the same iteration run unfused."""
    from megreader_amd.nn import BatchNorm2d, Conv2d
    from megreader_amd.nn.functional import ZeroArena
    mr.set_compute_dtype(dtype)
    torch.manual_seed(3)
    conv = Conv2d(64, 128, 3, padding=1, bias=True).to(DEV)
    bn = BatchNorm2d(128, fuse_relu=True).to(DEV)
    x0 = torch.randn(8, 64, 12, 20, device=DEV)
    gy = torch.randn(8, 128, 12, 20, device=DEV)

    def run(fused):
        conv.feeds_batch_norm = fused
        ZeroArena.reset(x0.device)
        bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
        for p in list(conv.parameters()) + list(bn.parameters()):
            p.grad = None
        x = x0.clone().requires_grad_(True)
        h = conv(x)
        assert (getattr(h, "_mr_bn_sums", None) is not None) == fused
        y = bn(h)
        y.backward(gy.to(y.dtype))
        return [t.detach().float().clone() for t in (y, x.grad, conv.weight.grad, conv.bias.grad, bn.weight.grad,
                                                     bn.bias.grad, bn.running_mean, bn.running_var)]
    a = run(False)
    assert conv.feeds_batch_norm                      # learned from the unfused forward
    b = run(True)
    tol = 2e-5 if dtype == torch.float32 else 5e-3    # bf16: one rounding of rstd-scaled values may flip
    for name, u, v in zip(("y", "dx", "dw", "db", "dgamma", "dbeta", "running_mean", "running_var"), a, b):
        # a bias in front of a BatchNorm has a mathematically zero gradient (round-off on both sides): measured against dw
        scale = float(a[2].abs().max()) if name == "db" else float(u.abs().max())
        err = float((u - v).abs().max()) / (scale + 1e-12)
        assert err < tol, (name, err)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("shape", [
    (512, 512, 4608, 0),      # dense, 64 tiles of 64x64 (the automatic choice for a launch this small)
    (2048, 256, 2304, 0),     # 128 tiles
    (1000, 136, 520, 0),      # ragged M / N, K not a multiple of 64 (last k-step predicated)
    (37, 64, 64, 0),          # a single k-step: fewer k-steps than stage buffers
    (200, 64, 192, 0),        # 3 k-steps
])
def test_nt_deep_pipeline_is_bit_identical(shape):
    """The 4-buffer main loop of the 4-wave NT kernel (3 k-steps of LDS-DMA in flight across raw barriers, mr_set_nt_deep)
    accumulates in the same order as the 2-buffer loop: dense GEMMs and convolutions (forward / dgrad, 3x3 and 1x1) must
    come out bit for bit the same, including k-step counts below the pipeline depth and a predicated last k-step."""
    from megreader_amd._lib import load
    lib = load()
    M, N, K, _ = shape
    dt = dtype_code(torch.bfloat16)
    g = torch.Generator().manual_seed(17)
    A = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    B = torch.randn(N, K, generator=g).to(DEV, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV)
    outs = []
    old = lib.mr_set_nt_deep(0)
    try:
        for mode in (0, 2):
            lib.mr_set_nt_deep(mode)
            C = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
            call("mr_gemm_nt", dt, ptr(A), K, ptr(B), K, ptr(C), N, ptr(bias), 1, M, N, K)
            outs.append(C)
        assert torch.equal(outs[0], outs[1])
        want = torch.relu(A.double().cpu() @ B.double().cpu().t() + bias.double().cpu())
        assert _rel_err(outs[1], want) < _tol(torch.bfloat16, K)
        # convolutions through the same kernel (AMODE 2): 3x3 pad 1 and 1x1 stride 2, forward and dgrad, + stats epilogue
        for (Nb, H, W, Cc, Kc, k, p, st) in [(2, 10, 12, 128, 64, 3, 1, 1), (3, 8, 16, 256, 128, 1, 0, 2)]:
            Ho, Wo = (H + 2 * p - k) // st + 1, (W + 2 * p - k) // st + 1
            x = torch.randn(Nb, H, W, Cc, generator=g).to(DEV, torch.bfloat16)
            w = (torch.randn(Kc, k, k, Cc, generator=g) * 0.05).to(DEV, torch.bfloat16)
            wt = (torch.randn(Cc, k, k, Kc, generator=g) * 0.05).to(DEV, torch.bfloat16)
            dy = torch.randn(Nb, Ho, Wo, Kc, generator=g).to(DEV, torch.bfloat16)
            res = []
            for mode in (0, 2):
                lib.mr_set_nt_deep(mode)
                y = torch.zeros(Nb, Ho, Wo, Kc, device=DEV, dtype=torch.bfloat16)
                sums = torch.zeros(lib.mr_bn_scratch_doubles(Kc), dtype=torch.float64, device=DEV)
                call("mr_conv2d_fwd_stats", dt, ptr(x), ptr(w), 0, ptr(y), ptr(sums), Nb, H, W, Cc, Cc, Kc, k, k, st, st, p,
                     p, 1, 1, Ho, Wo)
                dx = torch.zeros(Nb, H, W, Cc, device=DEV, dtype=torch.bfloat16)
                call("mr_conv2d_dgrad", dt, ptr(dy), ptr(wt), ptr(dx), Nb, H, W, Cc, Cc, Kc, Kc, k, k, st, st, p, p, 1, 1,
                     Ho, Wo)
                res.append((y, dx, sums[:16 * Kc].view(8, 2, Kc).sum(dim=0)))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
            assert float((res[0][2] - res[1][2]).abs().max()) <= 1e-9 * float(res[0][2].abs().max())
            want = TF.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), None, st,
                             p).permute(0, 2, 3, 1)
            assert _rel_err(res[1][0], want) < _tol(torch.bfloat16, Cc * k * k)
    finally:
        lib.mr_set_nt_deep(old)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(5, 4, 16, 128), (3, 7, 9, 72), (2, 16, 32, 64)])
def test_adaptive_avg_pool_multi_equals_separate_pools(dtype, shape):
    """The pyramid pooling module's four pools of one map as one autograd node (F.adaptive_avg_pool2d_multi): outputs equal the
    separate pools bit for bit (same summation order), the input gradient equals the sum of theirs (f32 accumulation, one
    rounding instead of four roundings + three adds), and both equal torch's f64 adaptive_avg_pool2d."""
    mr.set_compute_dtype(dtype)
    N, H, W, C = shape
    g = torch.Generator().manual_seed(23)
    x0 = torch.randn(N, C, H, W, generator=g).to(DEV)
    sizes = [1, 2, 3, 6]
    gys = [torch.randn(N, C, s_, s_, generator=g).to(DEV) for s_ in sizes]
    xa = x0.clone().requires_grad_(True)
    ya = [F.adaptive_avg_pool2d(xa, s_) for s_ in sizes]
    sum((y.float() * gy).sum() for y, gy in zip(ya, gys)).backward()
    xb = x0.clone().requires_grad_(True)
    yb = F.adaptive_avg_pool2d_multi(xb, sizes)
    sum((y.float() * gy).sum() for y, gy in zip(yb, gys)).backward()
    for u, v in zip(ya, yb):
        assert torch.equal(u, v)
    xr = x0.double().cpu().to(dtype).double().requires_grad_(True)      # the values the kernels see
    yr = [TF.adaptive_avg_pool2d(xr, s_) for s_ in sizes]
    sum((y * gy.to(dtype).double().cpu()).sum() for y, gy in zip(yr, gys)).backward()
    tol = 1e-5 if dtype == torch.float32 else 1.6e-2
    for v, r in zip(yb, yr):
        assert _rel_err(v, r) < tol
    assert _rel_err(xb.grad, xr.grad) < tol and _rel_err(xa.grad, xr.grad) < 2 * tol
    # only some branches used (unused outputs arrive as None gradients)
    xc = x0.clone().requires_grad_(True)
    yc = F.adaptive_avg_pool2d_multi(xc, sizes)
    (yc[1].float() * gys[1]).sum().backward()
    xd = x0.clone().requires_grad_(True)
    (F.adaptive_avg_pool2d(xd, 2).float() * gys[1]).sum().backward()
    assert _rel_err(xc.grad, xd.grad) < tol


@pytest.mark.parametrize("relu", [False, True])
def test_sync_batch_norm_two_virtual_ranks_equal_full_batch(relu):
    """apex.parallel.SyncBatchNorm (reference backbones/resnet.py:26-30 under config.sync_bn) on the HIP kernels: each of two
    virtual ranks holds half of a batch; the all-reduce hook adds the OTHER rank's contribution, computed independently in
    float64 torch from the data (forward: sum x, sum x^2, count; backward: sum g', sum g' xhat).  Every rank's output and input
    gradient must equal its half of plain BatchNorm on the full batch, the local dgamma / dbeta must add up to the full-batch
    ones, and the running statistics must be the full batch's."""
    from megreader_amd.apex.parallel import SyncBatchNorm
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(9)
    C, H, W = 64, 6, 10
    x = (torch.randn(8, C, H, W, device=DEV) * 1.5 + 0.3)
    gy = torch.randn(8, C, H, W, device=DEV)
    w0, b0 = (torch.rand(C, device=DEV) + 0.5), torch.randn(C, device=DEV) * 0.3
    # full-batch reference on the plain HIP BatchNorm
    xr = x.clone().requires_grad_(True)
    gam, bet = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yr = F.batch_norm(xr, gam, bet, rm, rv, True, 0.1, 1e-5, relu=relu)
    yr.backward(gy)
    # what each rank contributes to the two all-reduces, from the data in float64
    xd, gd = x.double(), gy.double()
    mean = xd.mean(dim=(0, 2, 3))
    var = xd.var(dim=(0, 2, 3), unbiased=False)
    xhat = (xd - mean.view(1, C, 1, 1)) / torch.sqrt(var.view(1, C, 1, 1) + 1e-5)
    gmask = gd * (yr.detach().double() > 0) if relu else gd
    halves = [slice(0, 3), slice(3, 8)]                       # uneven split: the count travels with the sums
    fwd_c = [torch.cat([xd[h].sum(dim=(0, 2, 3)), (xd[h] ** 2).sum(dim=(0, 2, 3)),
                        torch.tensor([float(xd[h].shape[0] * H * W)], dtype=torch.float64, device=DEV)]) for h in halves]
    bwd_c = [torch.cat([gmask[h].sum(dim=(0, 2, 3)), (gmask[h] * xhat[h]).sum(dim=(0, 2, 3))]) for h in halves]
    dg_sum, db_sum = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    for r, h in enumerate(halves):
        other = 1 - r
        calls = []

        def fake_all_reduce(t, other=other, calls=calls):
            calls.append(t.numel())
            t.add_(fwd_c[other] if t.numel() == 2 * C + 1 else bwd_c[other])
        bn = SyncBatchNorm(C, fuse_relu=relu).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(w0)
            bn.bias.copy_(b0)
        bn.all_reduce = fake_all_reduce
        xs = x[h].clone().requires_grad_(True)
        ys = bn(xs)
        ys.backward(gy[h])
        assert calls == [2 * C + 1, 2 * C]
        assert _rel_err(ys, yr[h]) < 2e-5, (r, _rel_err(ys, yr[h]))
        assert _rel_err(xs.grad, xr.grad[h]) < 5e-5, (r, _rel_err(xs.grad, xr.grad[h]))
        assert _rel_err(bn.running_mean, rm) < 1e-5 and _rel_err(bn.running_var, rv) < 1e-5
        assert int(bn.num_batches_tracked) == 1
        dg_sum += bn.weight.grad
        db_sum += bn.bias.grad
    assert _rel_err(dg_sum, gam.grad) < 2e-5 and _rel_err(db_sum, bet.grad) < 2e-5
    # eval mode and a single process: the plain BatchNorm2d path
    bn.all_reduce = None
    bn.eval()
    ye = bn(x)
    want = (x - bn.running_mean.view(1, C, 1, 1)) / torch.sqrt(bn.running_var.view(1, C, 1, 1) + 1e-5) * w0.view(1, C, 1, 1) \
        + b0.view(1, C, 1, 1)
    assert _rel_err(ye, torch.relu(want) if relu else want) < 2e-5


# ---------------------------------------------------------------------------------------------------------------
# mr_conv2d_dgrad_add: dx = dgrad(dy, w) + addend in the NT epilogue (the shortcut gradient of a ResNet block, reference
# backbones/resnet.py:152-181), on every NT kernel family a dgrad can be dispatched to.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,Cin,Cout,R", [
    (256, 8, 32, 256, 64, 1),      # ResNet layer1 conv1 dgrad at the benchmarked batch: M = 65536 (big / 128x128 tiles)
    (32, 4, 16, 1024, 256, 1),     # layer3: M = 2048, N = 1024 (small-M launch: deep-pipelined 4-wave kernel)
    (2, 20, 20, 2048, 512, 1),     # layer4 of the detector at batch 2: M = 800, ragged last tile
    (4, 9, 13, 64, 64, 3),         # 3x3 BasicBlock conv1, odd sizes (gathered operand, AMODE 2)
    (3, 7, 5, 72, 40, 3),          # Cin not a multiple of the k-step: register-staged / AMODE 3 path
])
def test_conv_dgrad_with_addend_epilogue(dtype, N, H, W, Cin, Cout, R):
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(N * H + Cin)
    pad = R // 2
    dy = torch.randn(N, H, W, Cout, generator=g).to(DEV, dtype)
    w = (torch.randn(Cout, Cin, R, R, generator=g) / math.sqrt(Cout * R * R)).to(DEV)
    add = torch.randn(N, H, W, Cin, generator=g).to(DEV, dtype)
    w_crsk = w.permute(1, 2, 3, 0).contiguous().to(dtype)            # [Cin][R][S][Cout]
    dt = dtype_code(dtype)
    plain = torch.empty(N, H, W, Cin, device=DEV, dtype=dtype)
    fused = torch.full((N, H, W, Cin), 3.0, device=DEV, dtype=dtype)
    args = (N, H, W, Cin, Cin, Cout, Cout, R, R, 1, 1, pad, pad, 1, 1, H, W)
    call("mr_conv2d_dgrad", dt, ptr(dy), ptr(w_crsk), ptr(plain), *args)
    call("mr_conv2d_dgrad_add", dt, ptr(dy), ptr(w_crsk), ptr(fused), ptr(add), *args)
    ref = TF.conv_transpose2d(dy.double().cpu().permute(0, 3, 1, 2), w_crsk.double().cpu().permute(3, 0, 1, 2), padding=pad)
    ref = ref.permute(0, 2, 3, 1) + add.double().cpu()
    assert _rel_err(fused, ref) < _tol(dtype, Cout * R * R)
    # against the unfused pair: same accumulator, one rounding less in bf16 -- equal up to one output ulp
    two = plain.float() + add.float()
    assert float((fused.float() - two).abs().max()) <= (1e-5 if dtype == torch.float32 else 2 ** -7) * float(two.abs().max())
    # in place: the addend may alias dx (each element is read before the same lane writes it)
    alias = add.clone()
    call("mr_conv2d_dgrad_add", dt, ptr(dy), ptr(w_crsk), ptr(alias), ptr(alias), *args)
    assert torch.equal(alias, fused)
    # null addend == plain dgrad, bit for bit
    again = torch.empty_like(plain)
    call("mr_conv2d_dgrad_add", dt, ptr(dy), ptr(w_crsk), ptr(again), 0, *args)
    assert torch.equal(again, plain)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("block", ["bottleneck", "basic"])
def test_residual_block_fork_equals_autograd_add(dtype, block):
    """Identity-shortcut blocks hand x to the shortcut as a second output of conv1's autograd node (nn.Conv2d.forward_fork), so
    the two gradients of x meet in ONE kernel.  Same block with the fork switched off (autograd's own add): same outputs,
    same gradients."""
    import copy
    from megreader_amd.backbones.resnet import BasicBlock, Bottleneck
    from megreader_amd.nn import modules as mrm
    mr.set_compute_dtype(dtype)
    torch.manual_seed(3)
    blk = (Bottleneck(256, 64) if block == "bottleneck" else BasicBlock(64, 64)).to(DEV).train()
    ref = copy.deepcopy(blk)
    c = 256 if block == "bottleneck" else 64
    x0 = torch.randn(4, c, 10, 14, device=DEV)
    gy = torch.randn(4, c, 10, 14, device=DEV)
    outs = []
    for m, fork in ((blk, True), (ref, False)):
        mrm.FORK_RESIDUAL = fork
        try:
            x = (x0 * 1.0).requires_grad_(True)
            xin = x * 1.0                                  # non-leaf input, like inside a network
            y = m(xin)
            y.float().backward(gy)
        finally:
            mrm.FORK_RESIDUAL = True
        outs.append((y.detach().float(), x.grad.detach().float(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
    (y1, dx1, g1), (y2, dx2, g2) = outs
    assert torch.equal(y1, y2)                             # the forward is the same kernels either way
    tol = 1e-5 if dtype == torch.float32 else 1.6e-2
    assert _rel_err(dx1, dx2) < tol
    for k in g1:
        assert _rel_err(g1[k], g2[k]) < tol, k
    # eval / no_grad: nothing to fuse, the module still returns the right thing
    blk.eval()
    with torch.no_grad():
        ye = blk(x0)
    assert ye.shape == y1.shape and torch.isfinite(ye.float()).all()


# ---------------------------------------------------------------------------------------------------------------
# BatchNorm-backward sums in the epilogue of the consuming convolution's dgrad (mr_conv2d_dgrad_bnb, F.BnBwdLink; VERDICT r3
# item 4, reference conv -> bn -> relu -> conv chains of backbones/resnet.py:113-181): same gradients as BatchNorm's own
# reduction pass, with and without a fused ReLU / residual / forked shortcut, and on the geometries that must fall back.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # (N, C_bn, H, W, Cout2, k2, stride2, relu, residual)
    (8, 64, 12, 20, 128, 3, 1, True, False),      # bn1 -> conv2 (3x3)
    (4, 256, 9, 7, 64, 1, 1, True, True),         # bn3 + residual + relu -> next block's 1x1 conv
    (16, 128, 8, 32, 512, 1, 1, False, False),    # no activation (CRNN: conv -> bn -> conv)
    (4, 64, 10, 10, 64, 3, 2, True, False),       # strided consumer: no epilogue for it, falls back to BatchNorm's own pass
])
def test_bn_backward_sums_in_dgrad_epilogue(dtype, case):
    from megreader_amd import nn as mnn
    N, C, H, W, Co, k, st, relu, with_res = case
    mr.set_compute_dtype(dtype)
    torch.manual_seed(C + H)
    conv0 = mnn.Conv2d(32, C, 3, padding=1, bias=False).to(DEV)
    bn = mnn.BatchNorm2d(C, fuse_relu=relu).to(DEV).train()
    conv = mnn.Conv2d(C, Co, k, stride=st, padding=k // 2, bias=False).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x0 = torch.randn(N, 32, H, W, device=DEV)
    res0 = torch.randn(N, C, H, W, device=DEV) if with_res else None
    Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
    gy = torch.randn(N, Co, Ho, Wo, device=DEV)
    outs = []
    default = F.BNB_EPILOGUE
    for fused in (True, False):
        F.BNB_EPILOGUE = fused
        conv.sole_consumer_of_bn = True
        try:
            for p in list(conv0.parameters()) + list(bn.parameters()) + list(conv.parameters()):
                p.grad = None
            x = x0.clone().requires_grad_(True)
            res = res0.clone().requires_grad_(True) if with_res else None
            h = conv0(x)
            hb = bn(h, residual=res) if with_res else bn(h)
            assert (getattr(hb, "_mr_bnb_link", None) is not None) == fused
            y = conv(hb)
            y.float().backward(gy)
        finally:
            F.BNB_EPILOGUE = default
        outs.append((y.detach().float(), x.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                     conv0.weight.grad.clone(), res.grad.float() if with_res else None))
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2
    # the forward is the same code either way; the batch statistics come from f64 atomics whose order is not fixed
    assert _rel_err(outs[0][0], outs[1][0]) < 1e-6, _rel_err(outs[0][0], outs[1][0])
    for a, b, name in zip(outs[0][1:], outs[1][1:], ("dx", "dgamma", "dbeta", "dw0", "dres")):
        if a is not None:
            assert _rel_err(a, b) < tol, (name, _rel_err(a, b))


# ---------------------------------------------------------------------------------------------------------------
# Data gradient of a strided 1x1 convolution (ResNet downsample branch, backbones/resnet.py:204-213) as the dense dgrad on the
# sub-sampled grid + mr_scatter_strided: equal to the implicit-GEMM dgrad over every output pixel (same products, same order
# per pixel) and to torch; odd sizes, anisotropic strides, the sampled grid not reaching the last row / column.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C,K,H,W,stride", [(4, 64, 128, 8, 32, (2, 2)), (3, 128, 64, 7, 9, (2, 2)), (2, 256, 512, 5, 16, (2, 1)),
                                             (2, 64, 64, 6, 10, (3, 2)), (32, 1024, 2048, 4, 16, (2, 2))])
def test_strided_pointwise_dgrad_is_dense_dgrad_plus_scatter(dtype, N, C, K, H, W, stride):
    from megreader_amd import nn as mnn
    mr.set_compute_dtype(dtype)
    torch.manual_seed(C + H)
    conv = mnn.Conv2d(C, K, 1, stride=stride, bias=False).to(DEV)
    x0 = torch.randn(N, C, H, W, device=DEV)
    Ho, Wo = (H - 1) // stride[0] + 1, (W - 1) // stride[1] + 1
    gy = torch.randn(N, K, Ho, Wo, device=DEV)
    outs = []
    default = F.POINTWISE_STRIDED_DGRAD
    for fast in (True, False):
        F.POINTWISE_STRIDED_DGRAD = fast
        try:
            x = x0.clone().requires_grad_(True)
            conv.weight.grad = None
            y = conv(x)
            y.float().backward(gy)
            outs.append((x.grad.float(), conv.weight.grad.clone()))
        finally:
            F.POINTWISE_STRIDED_DGRAD = default
    assert _rel_err(outs[0][0], outs[1][0]) < (1e-6 if dtype == torch.float32 else 4e-3)
    assert _rel_err(outs[0][1], outs[1][1]) < 1e-5          # the weight gradient is untouched (f32 atomics: order not fixed)
    xr = x0.to(dtype).double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, conv.weight.detach().to(dtype).double(), None, stride)
    yr.backward(gy.to(dtype).double())
    assert _rel_err(outs[0][0], xr.grad) < (1e-5 if dtype == torch.float32 else 1.6e-2)
    # zeros exactly where no output pixel samples the input
    mask = torch.zeros(H, W, dtype=torch.bool, device=DEV)
    mask[::stride[0], ::stride[1]] = True
    assert bool((outs[0][0][:, :, ~mask] == 0).all())
