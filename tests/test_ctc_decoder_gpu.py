"""decoders.CTCDecoder (reference decoders/ctc_decoder.py:13-66) on HIP against its oracle (oracle/ctc_decoder.py, pinned
bit-identical to the unmodified reference on CPU by tests/test_oracle_models.py): fp32 loss / log-probabilities / every
parameter and input gradient, eval softmax; bf16 runs close."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.decoders import CTCDecoder  # noqa: E402
from oracle.ctc_decoder import CTCDecoderOracle  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _setup(n=6, cin=64, inner=64, h=16, w=64):
    torch.manual_seed(11)
    ora = CTCDecoderOracle(cin, inner_channels=inner).train()
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(n, cin, h, w, generator=g)
    lengths = torch.randint(2, 9, (n,), generator=g)
    labels = torch.zeros((n, 32), dtype=torch.long)
    for i in range(n):
        labels[i, :lengths[i]] = torch.randint(1, 38, (int(lengths[i]),), generator=g)
    return ora, feat, labels, lengths


def test_fp32_parity_vs_oracle():
    mr.set_compute_dtype(torch.float32)
    ora, feat, labels, lengths = _setup()
    model = CTCDecoder(in_channels=64, inner_channels=64)
    model.load_state_dict(ora.state_dict(), strict=True)
    model.to(DEV).train()
    ora64 = copy.deepcopy(ora).double()
    x64 = feat.double().requires_grad_(True)
    l64, p64 = ora64(x64, targets=labels, lengths=lengths, train=True)
    l64.backward()
    xo = feat.clone().requires_grad_(True)
    lo, po = ora(xo, targets=labels, lengths=lengths, train=True)
    lo.backward()
    xd = feat.to(DEV).requires_grad_(True)
    loss, pred = model(xd, targets=labels.to(DEV), lengths=lengths.to(DEV), train=True)
    assert loss.dim() == 0 and loss.dtype == torch.float32 and pred.shape == po.shape and pred.dtype == torch.float32
    assert abs(float(loss) - float(lo)) < 1e-4 * max(1.0, abs(float(lo)))
    assert float((pred.cpu() - po).abs().max()) < 1e-4
    loss.backward()
    g64 = dict(ora64.named_parameters())
    g32 = dict(ora.named_parameters())
    worst = 0.0
    for k, p in model.named_parameters():
        scale = float(g64[k].grad.abs().max())
        if scale < 1e-9:      # conv biases in front of a BatchNorm: zero gradient
            continue
        e_hip, e_cpu = _rel(p.grad, g64[k].grad), _rel(g32[k].grad, g64[k].grad)
        worst = max(worst, e_hip)
        assert e_hip < max(4 * e_cpu, 1e-3), (k, e_hip, e_cpu)
    assert _rel(xd.grad, x64.grad) < max(4 * _rel(xo.grad, x64.grad), 1e-3)
    print("CTCDecoder fp32: loss |d| %.2e, log-prob max|d| %.2e, worst gradient error vs f64 %.2e" %
          (abs(float(loss) - float(lo)), float((pred.cpu() - po).abs().max()), worst))
    ora.eval()
    model.eval()
    with torch.no_grad():
        ev, evo = model(feat.to(DEV), train=False), ora(feat, train=False)
    assert ev.shape == evo.shape and float((ev.cpu() - evo).abs().max()) < 1e-5
    assert torch.equal(ev.cpu().argmax(dim=1), evo.argmax(dim=1))


def test_bf16_runs_close():
    mr.set_compute_dtype(torch.bfloat16)
    ora, feat, labels, lengths = _setup()
    model = CTCDecoder(in_channels=64, inner_channels=64)
    model.load_state_dict(ora.state_dict(), strict=True)
    model.to(DEV).train()
    lo, _ = ora(feat, targets=labels, lengths=lengths, train=True)
    loss, _ = model(feat.to(DEV), targets=labels.to(DEV), lengths=lengths.to(DEV), train=True)
    loss.backward()
    assert abs(float(loss) - float(lo)) < 5e-2 * max(1.0, abs(float(lo)))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
