"""Fused backbone stem (Conv2d(Cin->64,3x3,p1) + ReLU + MaxPool2d(2,2), megreader_amd/csrc/stem.hip) against a torch
CPU reference of the three reference ops (backbones/crnn.py:17-19,48-55) and against the unfused HIP path; plus the
batched weight-prep / multi-segment accumulate entry points the training step uses around it."""
import ctypes

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd._lib import call, ptr  # noqa: E402
from megreader_amd.nn import functional as F  # noqa: E402
from megreader_amd.nn import prep  # noqa: E402
from megreader_amd.backbones import crnn_backbone  # noqa: E402
from megreader_amd.optim import FusedAdam  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


def _reference(x, w, b, gy, round_bf16):
    """CPU torch: conv -> relu -> maxpool, gradients wrt w and b for upstream gradient gy (pooled NCHW)."""
    if round_bf16:
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    x = x.double()
    w = w.double().requires_grad_(True)
    b = b.double().requires_grad_(True)
    y = TF.max_pool2d(TF.relu(TF.conv2d(x, w, b, padding=1)), 2, 2)
    y.backward(gy.double())
    return y.detach(), w.grad, b.grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(3, 3, 32, 128), (2, 1, 8, 20), (5, 3, 2, 2), (2, 3, 16, 250), (2, 1, 4, 64),
                                     (1, 3, 6, 32)])
@pytest.mark.parametrize("channels_last_weight", [False, True])
def test_stem_matches_reference_ops(dtype, N, C, H, W, channels_last_weight):
    mr.set_compute_dtype(dtype)
    torch.manual_seed(N * 100 + W)
    # small dyadic values: every product and partial sum is exact in f32 (and in bf16 operands), so the arg-max of
    # each pooling window -- including exact ties, resolved "first maximum" by all implementations -- is the same
    # in the f64 reference and on the GPU.  (With random reals a handful of near-ties flip between f32 and f64
    # accumulation orders and move whole gradient contributions.)
    x = torch.randint(-3, 4, (N, C, H, W)).float()
    w = torch.randint(-4, 5, (64, C, 3, 3)).float() / 8
    b = torch.randint(-8, 9, (64,)).float() / 8
    gy = torch.randint(-4, 5, (N, 64, H // 2, W // 2)).float()
    y_ref, dw_ref, db_ref = _reference(x, w, b, gy, dtype == torch.bfloat16)

    wd = w.to(DEV)
    if channels_last_weight:
        wd = wd.contiguous(memory_format=torch.channels_last)
    wd.requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    y = F.stem_conv_relu_pool(x.to(DEV), wd, bd)
    assert y.shape == (N, 64, H // 2, W // 2) and y.dtype == dtype
    y.backward(gy.to(DEV).to(dtype))
    if dtype == torch.float32:
        assert torch.equal(y.double().cpu(), y_ref)
    else:  # outputs are the exact values rounded to bf16
        assert torch.equal(y.float().cpu(), y_ref.float().bfloat16().float())
    # gradients are sums of small integers times dyadic inputs: exact in f32 whatever the summation order
    assert torch.equal(wd.grad.double().cpu(), dw_ref)
    assert torch.equal(bd.grad.double().cpu(), db_ref)


def test_stem_fp32_equals_unfused_hip_path():
    """The CRNN backbone's first stage through the fused kernel vs the generic Conv2d -> MaxPool2d kernels."""
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(3)
    net = crnn_backbone().to(DEV).train()
    stem = net.cnn[0]
    conv = stem[0][0]
    with torch.no_grad():  # exact (dyadic) data: no near-tie arg-max flips between the two accumulation orders
        conv.weight.copy_(torch.randint(-4, 5, conv.weight.shape, device=DEV).float() / 8)
        conv.bias.copy_(torch.randint(-8, 9, conv.bias.shape, device=DEV).float() / 8)
    x = torch.randint(-3, 4, (4, 3, 32, 128), device=DEV).float()
    y_f = stem(x)
    g = torch.randint(-4, 5, tuple(y_f.shape), device=DEV).float().contiguous(memory_format=torch.channels_last)
    y_f.backward(g)
    gw_f, gb_f = conv.weight.grad.clone(), conv.bias.grad.clone()
    conv.weight.grad = None
    conv.bias.grad = None
    y_u = torch.nn.Sequential.forward(stem, x)  # the generic Conv2d -> (ReLU) -> MaxPool2d kernels
    y_u.backward(g)
    assert torch.equal(y_f, y_u)
    assert torch.equal(gw_f, conv.weight.grad)
    assert torch.equal(gb_f, conv.bias.grad)


def test_accumulate_multi():
    torch.manual_seed(0)
    sizes = [1, 7, 64, 1000, 4096, 12345, 3, 2, 77, 100000]   # > 8 segments: two launches
    dst = [torch.randn(n, device=DEV) for n in sizes]
    src = [torch.randn(n, device=DEV) for n in sizes]
    want = [d + s for d, s in zip(dst, src)]
    F.accumulate_multi(list(zip(dst, src)))
    for d, w in zip(dst, want):
        assert torch.equal(d, w)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_prep_batch_equals_individual_prep(dtype):
    """mr_prep_batch regenerates exactly what the per-layer prep kernels produce, and the prep cache refreshes
    after a fused-optimizer update."""
    mr.set_compute_dtype(dtype)
    torch.manual_seed(1)
    from megreader_amd.nn import Conv2d, Linear, LSTM

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = Conv2d(8, 16, 3, 1, 1)
            self.c2 = Conv2d(16, 20, (3, 2), 1, (1, 0))   # padded output channels (20 -> 24 in bf16)
            self.rnn = LSTM(16, 8, bidirectional=True)
            self.fc = Linear(16, 10)

        def forward(self, x):
            y = self.c2(self.c1(x))[:, :16]           # [N,16,H,W-1]
            seq = y.float().mean(2).permute(2, 0, 1)  # [T,N,16]
            out, _ = self.rnn(seq)
            return self.fc(out).float().sum()

    net = Net().to(DEV)
    opt = FusedAdam(net.parameters(), lr=1e-2)
    x = torch.randn(2, 8, 4, 9, device=DEV, requires_grad=True)
    for _ in range(3):
        opt.zero_grad()
        net(x).backward()
        opt.step()
    entries = [e for p in net.parameters() for e in p.__dict__.get("_mr_prep", {}).values()]
    assert len(entries) >= 4
    cached = [[b.clone() if b is not None else None for b in e.buffers] for e in entries]
    # rebuild every image through the individual entry points into fresh buffers and compare bit for bit
    prep.invalidate(net)
    opt.zero_grad()
    net(x).backward()
    fresh = [e for p in net.parameters() for e in p.__dict__.get("_mr_prep", {}).values()]
    assert len(fresh) == len(entries)
    for old, e in zip(cached, fresh):
        for a, b in zip(old, e.buffers):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b)
