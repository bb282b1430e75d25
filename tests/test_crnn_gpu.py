"""End-to-end parity of the HIP CRNN training path against the oracle / the committed golden vectors.

The golden file was produced by the UNMODIFIED reference (oracle/gen_golden.py); the oracle is bit-identical to
it on CPU.  Bars (BASELINE.json north_star): loss and log-probabilities within 1e-4 in fp32, greedy decode
bit-exact.  bf16 is the benchmark dtype: its drift is reported and bounded loosely.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.backbones import crnn_backbone  # noqa: E402
from megreader_amd.decoders import CRNNDecoder  # noqa: E402
from megreader_amd.optim import FusedAdam  # noqa: E402
from oracle.crnn import CRNNOracle, synthetic_batch  # noqa: E402
from oracle.decode import greedy_decode  # noqa: E402

DEV = "cuda"


class BasicModel(torch.nn.Module):
    """reference structure/model.py:16-24: decoder(backbone(data), *args, **kwargs)."""

    def __init__(self):
        super().__init__()
        self.backbone = crnn_backbone()
        self.decoder = CRNNDecoder(in_channels=512, inner_channels=256, need_reduce=False)

    def forward(self, data, *args, **kwargs):
        return self.decoder(self.backbone(data), *args, **kwargs)


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return torch.load(os.path.join(golden_dir, "crnn_golden.pt"), weights_only=False)


def _oracle(golden):
    torch.manual_seed(golden['weight_seed'])
    return CRNNOracle()


def _to_dev(batch):
    return batch['image'].to(DEV), batch['label'].to(DEV), batch['length'].to(DEV).long()


def test_state_dict_interchange(golden):
    model = BasicModel()
    ora = _oracle(golden)
    assert list(model.state_dict().keys()) == golden['state_keys']
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == golden['state_shapes'][k], k
    model.load_state_dict(ora.state_dict(), strict=True)
    model.to(DEV)
    for k, v in model.state_dict().items():
        s, a = golden['state_checksums'][k]
        assert abs(float(v.double().sum()) - s) <= 1e-6 * max(1.0, a), k


def test_fp32_training_parity_vs_reference_golden(golden):
    mr.set_compute_dtype(torch.float32)
    ora = _oracle(golden)
    model = BasicModel()
    model.load_state_dict(ora.state_dict())
    model.to(DEV).train()
    img, lab, ln = _to_dev(golden['batch'])
    loss, pred = model(img, targets=lab, lengths=ln, train=True)
    assert loss.dtype == torch.float64 and pred.dtype == torch.float64
    assert abs(float(loss) - float(golden['train_loss'])) < 1e-4
    assert float((pred.cpu() - golden['train_log_probs']).abs().max()) < 1e-4
    loss.mean().backward()
    worst = 0.0
    for k, p in model.named_parameters():
        norm, head = golden['grad_stats'][k]
        g = p.grad.float().cpu()
        # conv biases in front of a BatchNorm have a mathematically zero gradient (round-off only): floor the scale
        scale = max(norm, 1e-4)
        rel = abs(float(g.double().norm()) - norm) / scale
        worst = max(worst, rel)
        assert rel < 2e-3, (k, rel, norm)
        assert float((g.flatten()[:8] - head).abs().max()) < 2e-3 * max(float(head.abs().max()), scale), k
    print("worst relative grad-norm error (fp32):", worst)
    for k, v in model.state_dict().items():
        if 'running' in k:
            assert float((v.cpu() - golden['bn_after'][k]).abs().max()) < 1e-4, k


def test_fp32_adam_trajectory(golden):
    mr.set_compute_dtype(torch.float32)
    ora = _oracle(golden)
    model = BasicModel()
    model.load_state_dict(ora.state_dict())
    model.to(DEV).train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    img, lab, ln = _to_dev(golden['batch'])
    # the golden trajectory starts after one plain forward/backward that only moved BN running stats
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss, _ = model(img, targets=lab, lengths=ln, train=True)
        loss.mean().backward()
        opt.step()
        losses.append(float(loss))
    for a, b in zip(losses, golden['adam_losses']):
        assert abs(a - b) < 2e-3 * max(1.0, abs(b)), (losses, golden['adam_losses'])


def test_gradient_sinks_match_autograd_accumulation(golden):
    """Weight gradients accumulated straight into the flat optimizer buffer (megreader_amd.nn.functional.grad_sink)
    equal the ones autograd accumulates, including over two backward passes, and revert to the autograd path after
    `model.zero_grad()` (grad = None)."""
    mr.set_compute_dtype(torch.float32)
    ora = _oracle(golden)
    img, lab, ln = _to_dev(golden['batch'])

    def grads(use_sink, passes):
        torch.manual_seed(0)
        model = BasicModel()
        model.load_state_dict(ora.state_dict())
        model.to(DEV).train()
        if use_sink:
            opt = FusedAdam(model.parameters(), lr=1e-3)
            opt.zero_grad()
            assert any(getattr(p, "_mr_grad_sink", None) is not None for p in model.parameters())
        for _ in range(passes):
            loss, _ = model(img, targets=lab, lengths=ln, train=True)
            loss.mean().backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}, model

    for passes in (1, 2):
        ref, _ = grads(False, passes)
        got, model = grads(True, passes)
        for k in ref:
            scale = float(ref[k].abs().max()) + 1e-6
            assert float((ref[k] - got[k]).abs().max()) <= 2e-5 * scale + 1e-7, (passes, k)
    # after model.zero_grad() the sinks are inactive: plain autograd grads, flat buffer untouched
    model.zero_grad(set_to_none=True)
    loss, _ = model(img, targets=lab, lengths=ln, train=True)
    loss.mean().backward()
    ref1, _ = grads(False, 1)
    for k, p in model.named_parameters():
        scale = float(ref1[k].abs().max()) + 1e-6
        assert float((ref1[k] - p.grad).abs().max()) <= 5e-4 * scale + 1e-7, k


def test_eval_and_greedy_decode_bit_exact(golden):
    mr.set_compute_dtype(torch.float32)
    ora = _oracle(golden)
    # golden eval ran after BN stats moved by 4 training forwards and 3 Adam steps; redo them on the oracle (CPU)
    batch = golden['batch']
    ora.train()
    l, _ = ora(batch['image'], targets=batch['label'], lengths=batch['length'].long(), train=True)
    opt = torch.optim.Adam(ora.parameters(), lr=1e-3)
    for _ in range(3):
        opt.zero_grad()
        l, _ = ora(batch['image'], targets=batch['label'], lengths=batch['length'].long(), train=True)
        l.mean().backward()
        opt.step()
    ora.eval()
    with torch.no_grad():
        ev_o = ora(batch['image'], train=False)
    # same torch CPU kernels as the reference; bit-exact in the build container, round-off level on other hosts
    assert float((ev_o - golden['eval_pred']).abs().max()) < 5e-5
    model = BasicModel()
    model.load_state_dict(ora.state_dict())
    model.to(DEV).eval()
    with torch.no_grad():
        ev = model(batch['image'].to(DEV), train=False)
    assert ev.shape == golden['eval_pred'].shape
    assert float((ev.cpu() - ev_o).abs().max()) < 1e-4
    assert float((ev.cpu() - golden['eval_pred']).abs().max()) < 1.5e-4
    dec = greedy_decode(ev.cpu().numpy())
    top2 = golden['eval_pred'].topk(2, dim=1).values
    print("min top-1/top-2 margin:", float((top2[:, 0] - top2[:, 1]).min()))
    assert (torch.from_numpy(dec) == golden['eval_decode']).all()


def test_bf16_training_close_to_oracle(golden):
    mr.set_compute_dtype(torch.bfloat16)
    ora = _oracle(golden)
    model = BasicModel()
    model.load_state_dict(ora.state_dict())
    model.to(DEV).train()
    img, lab, ln = _to_dev(golden['batch'])
    loss, pred = model(img, targets=lab, lengths=ln, train=True)
    loss.mean().backward()
    drift = abs(float(loss) - float(golden['train_loss']))
    print("bf16 loss drift vs reference:", drift)
    assert drift < 0.1
    for k, p in model.named_parameters():
        norm, _ = golden['grad_stats'][k]
        if norm < 1e-5:
            continue  # conv bias feeding a BatchNorm: mathematically zero gradient, only round-off
        rel = abs(float(p.grad.double().norm()) - norm) / norm
        assert rel < 0.25, (k, rel, norm)


def test_full_size_batch_runs_and_learns():
    """BASELINE config 2 shape (N=256, 32x128, bf16): loss finite and decreasing over a few fused-Adam steps."""
    mr.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    model = BasicModel().to(DEV).train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    batch = synthetic_batch(256, 32, 128, seed=0)
    img, lab, ln = _to_dev(batch)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss, _ = model(img, targets=lab, lengths=ln, train=True)
        loss.mean().backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("reduce_func,loss_func", [("conv", "pytorch"), ("pooling", "custom"), ("conv", "custom")])
def test_decoder_variants_need_reduce_and_python_ctc(reduce_func, loss_func):
    """CRNNDecoder(need_reduce=True, reduce_func='conv'|'pooling') and loss_func != 'pytorch' (reference decoders/
    crnn.py:36-50, decoders/ctc_loss.py) against oracle/crnn.py:CRNNDecoderOracle, which tests/test_oracle_models.py pins
    to the unmodified reference.  fp32: loss / log-probabilities 1e-4, every gradient within 2e-3 of its max."""
    from oracle.crnn import CRNNDecoderOracle
    mr.set_compute_dtype(torch.float32)
    try:
        cin = 48 if reduce_func == "conv" else 32
        kw = dict(inner_channels=32, in_channels=cin, need_reduce=True, reduce_func=reduce_func, loss_func=loss_func)
        torch.manual_seed(5)
        ora = CRNNDecoderOracle(num_classes=38, **kw).train()
        model = CRNNDecoder(**kw)
        model.load_state_dict(ora.state_dict(), strict=True)
        model.to("cuda").train()
        g = torch.Generator().manual_seed(1)
        feat = torch.randn(4, cin, 8, 20, generator=g)
        labels = torch.randint(2, 38, (4, 6), generator=g, dtype=torch.int32)
        lengths = torch.tensor([6, 3, 4, 1], dtype=torch.int32)
        fo = feat.clone().requires_grad_(True)
        lo, po = ora(fo, targets=labels, lengths=lengths, train=True)
        lo.mean().backward()
        fd = feat.to("cuda").requires_grad_(True)
        ld, pd = model(fd, targets=labels.to("cuda"), lengths=lengths.to("cuda"), train=True)
        assert ld.shape == lo.shape and ld.dtype == lo.dtype == torch.float64
        assert (ld.cpu() - lo).abs().max() < 1e-4 * max(1.0, float(lo.abs().max()))
        assert (pd.cpu() - po).abs().max() < 1e-4
        ld.mean().backward()

        def rel(a, b):
            return float((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))
        assert rel(fd.grad, fo.grad) < 2e-3
        op = dict(ora.named_parameters())
        for k, p in model.named_parameters():
            assert p.grad is not None, k
            if ".0.bias" in k and "fpn2rnn" in k:     # conv bias in front of a BatchNorm: mathematically zero gradient
                continue
            assert rel(p.grad, op[k].grad) < 2e-3, (k, rel(p.grad, op[k].grad))
        model.eval(), ora.eval()
        with torch.no_grad():
            assert (model(fd).cpu() - ora(feat)).abs().max() < 1e-4
    finally:
        mr.set_compute_dtype(torch.bfloat16)
