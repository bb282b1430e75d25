"""ResNet50-FPN backbone + attention (GRU) decoder on HIP vs golden vectors produced by the unmodified reference
(oracle/gen_golden.py fpn_attention; gt_as_output=True, resnet_pretrained=False)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.backbones import Resnet50FPN  # noqa: E402
from megreader_amd.decoders import AttentionDecoder  # noqa: E402
from oracle.fpn_attention import FPNAttentionOracle  # noqa: E402

DEV = "cuda"


class BasicModel(torch.nn.Module):  # reference structure/model.py:16-24
    def __init__(self):
        super().__init__()
        self.backbone = Resnet50FPN(resnet_pretrained=False)
        self.decoder = AttentionDecoder(in_channels=256, gt_as_output=True)

    def forward(self, data, *args, **kwargs):
        return self.decoder(self.backbone(data), *args, **kwargs)


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return torch.load(os.path.join(golden_dir, "fpn_attention_golden.pt"), weights_only=False)


def _models(golden, dtype):
    mr.set_compute_dtype(dtype)
    torch.manual_seed(golden['weight_seed'])
    ora = FPNAttentionOracle()
    model = BasicModel()
    model.load_state_dict(ora.state_dict(), strict=True)
    return ora, model.to(DEV)


def test_state_dict_mirrors_reference(golden):
    torch.manual_seed(golden['weight_seed'])
    model = BasicModel()
    assert list(model.state_dict().keys()) == golden['state_keys']
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == golden['state_shapes'][k], k
        s, a = golden['state_checksums'][k]
        assert abs(float(v.double().sum()) - s) <= 1e-6 * max(1.0, a), k


def test_fp32_parity_vs_reference_golden(golden):
    ora, model = _models(golden, torch.float32)
    b = golden['batch']
    model.train()
    loss, att = model(b['image'].to(DEV), targets=b['label'].to(DEV), lengths=b['length'].to(DEV).long(), train=True)
    assert loss.shape == golden['train_loss'].shape and att.shape == golden['train_attention'].shape
    assert float((loss.cpu() - golden['train_loss']).abs().max()) < 1e-4 * float(golden['train_loss'].abs().max())
    assert float((att.cpu() - golden['train_attention']).abs().max()) < 1e-4
    loss.mean().backward()
    worst = 0.0
    for k, p in model.named_parameters():
        gs = golden['grad_stats'][k]
        if gs is None:
            assert p.grad is None, k          # unused fc / smooth of the plain ResNet receive no gradient
            continue
        norm, head = gs
        if norm < 1e-6:
            continue
        assert p.grad is not None, k
        rel = abs(float(p.grad.double().norm()) - norm) / norm
        worst = max(worst, rel)
        assert rel < 2e-2, (k, rel, norm)
    print("worst relative grad-norm error:", worst)
    model.eval()
    with torch.no_grad():
        pred = model(b['image'].to(DEV), train=False)
    assert pred.dtype == torch.int32 and (pred.cpu() == golden['eval_pred']).all()   # greedy decode bit-exact


def test_bf16_runs_close(golden):
    ora, model = _models(golden, torch.bfloat16)
    b = golden['batch']
    model.train()
    loss, _ = model(b['image'].to(DEV), targets=b['label'].to(DEV), lengths=b['length'].to(DEV).long(), train=True)
    loss.mean().backward()
    rel = float(((loss.cpu() - golden['train_loss']).abs() / golden['train_loss'].abs()).max())
    print("bf16 relative loss drift:", rel)
    assert rel < 0.05


# ---------------------------------------------------------------------------------------------------------------
# Scheduled sampling (gt_as_output = None, what experiments/recognition/fpn50-attention-decoder.yaml trains with): one
# teacher-forcing coin per decode step, reference decoders/attention_decoder.py:107-110.  The coins live in DEVICE memory and
# the step kernels select target / arg-max feedback from them (a host coin would be frozen into a captured step).
# ---------------------------------------------------------------------------------------------------------------
def _decoder_pair():
    from oracle.fpn_attention import AttentionDecoderOracle
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(11)
    ora = AttentionDecoderOracle(256, 38).train()
    dec = AttentionDecoder(in_channels=256)                 # gt_as_output=None: np.random coins, like the reference
    dec.load_state_dict(ora.state_dict(), strict=True)
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(6, 256, 16, 64, generator=g)
    lab = torch.randint(2, 38, (6, 32), generator=g, dtype=torch.int32)
    ln = torch.randint(3, 11, (6,), generator=g)
    return ora, dec.to(DEV).train(), feat, lab, ln


def test_scheduled_sampling_coins_follow_the_reference_order():
    """Eager step: the coins are np.random draws in the reference's order; the loop then equals the oracle run with the same
    coin list (arg-max feedback where the coin says so), loss / attention maps / every decoder gradient."""
    import numpy as np
    ora, dec, feat, lab, ln = _decoder_pair()
    np.random.seed(123)
    coins = [bool(np.random.rand() < 0.5) for _ in range(32)]
    assert 8 < sum(coins) < 24
    loss_o, att_o = ora(feat, targets=lab, lengths=ln, train=True, coins=coins)
    loss_o.mean().backward()
    np.random.seed(123)                                      # the HIP module draws the same 32 coins
    x = feat.to(DEV)
    loss, att = dec(x, targets=lab.to(DEV), lengths=ln.to(DEV), train=True)
    loss.mean().backward()
    assert float((loss.cpu() - loss_o).abs().max()) < 2e-4 * float(loss_o.abs().max())
    assert float((att.cpu() - att_o).abs().max()) < 1e-4
    po = dict(ora.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in po.values() if p.grad is not None)
    worst_dec, worst_enc = 0.0, 0.0
    for k, p in dec.named_parameters():
        if po[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        scale = float(po[k].grad.abs().max())
        if scale < 1e-5 * gmax:
            continue                  # conv biases in front of a BatchNorm: mathematically zero gradient, round-off on both sides
        e = float((p.grad.cpu() - po[k].grad).abs().max()) / scale
        if k.startswith("encode."):
            worst_enc = max(worst_enc, e)
        else:
            worst_dec = max(worst_dec, e)
    print("scheduled sampling, %d of 32 steps teacher-forced: worst gradient difference vs the oracle %.2e of max|g| on the "
          "decode-loop parameters, %.2e on the conv encoder (batch statistics over 6 samples)" %
          (sum(coins), worst_dec, worst_enc))
    assert worst_dec < 2e-3 and worst_enc < 1e-1
    # all-False coins == pure arg-max feedback == gt_as_output=False
    dec.gt_as_output = False
    loss_f, _ = dec(x, targets=lab.to(DEV), lengths=ln.to(DEV), train=True)
    loss_of, _ = ora(feat, targets=lab, lengths=ln, train=True, coins=[False] * 32)
    assert float((loss_f.cpu() - loss_of).abs().max()) < 2e-4 * float(loss_of.abs().max())


def test_scheduled_sampling_is_redrawn_on_every_graph_replay():
    """Captured step (the graphed drop-in trainer): the coins come from torch's device generator inside the graph, so every
    replay draws a new pattern.  With lr = 0 the weights never move: a frozen pattern would give the same loss on every
    replay; redrawn coins change which words are fed back and therefore the loss."""
    from megreader_amd.optim import FusedAdam
    from megreader_amd.runtime import GraphedTrainStep
    _ora, dec, feat, lab, ln = _decoder_pair()
    x, lab_d, ln_d = feat.to(DEV), lab.to(DEV), ln.to(DEV)
    opt = FusedAdam(dec.parameters(), lr=0.0)
    opt.zero_grad()

    def loss_fn():
        return dec(x, targets=lab_d, lengths=ln_d, train=True)[0].mean()
    step = GraphedTrainStep(loss_fn, opt, [], warmup=2)
    losses = []
    for _ in range(8):
        losses.append(round(float(step()), 5))
    print("losses of 8 replays with redrawn teacher-forcing coins:", losses)
    assert len(set(losses)) >= 4, losses
    del step                                                  # one live captured step per model (runtime.GraphedTrainStep)
    dec.gt_as_output = True                                   # fixed teacher forcing: a new capture, identical replays
    step2 = GraphedTrainStep(loss_fn, opt, [], warmup=1)
    fixed = [round(float(step2()), 5) for _ in range(3)]
    assert len(set(fixed)) == 1, fixed


def test_two_live_captured_steps_replay_consistently():
    """VERDICT r4 parity item 4.  Two GraphedTrainStep captures of ONE model alive at the same time -- the older one still
    holding its loss WITH its autograd graph, which keeps the parameters' AccumulateGrad nodes (and the stream they are bound
    to) alive.  With a capture stream per instance the newer graph's replays raced (partial decode-step sums in round 4, a memory
    aperture violation after some tens of replays in round 5: profiles/r05_diag_two_live_captures.txt); all instances now
    capture on one stream per device (runtime._CAPTURE_STREAMS).  lr = 0 and fixed teacher forcing: every replay of the second
    graph must give the same loss, equal to an eager step's, and interleaved replays of the first graph must not disturb it."""
    import warnings
    from megreader_amd.optim import FusedAdam
    from megreader_amd.runtime import GraphedTrainStep
    _ora, dec, feat, lab, ln = _decoder_pair()
    x, lab_d, ln_d = feat.to(DEV), lab.to(DEV), ln.to(DEV)
    opt = FusedAdam(dec.parameters(), lr=0.0)
    opt.zero_grad()
    hold = {}

    def loss_fn():
        loss = dec(x, targets=lab_d, lengths=ln_d, train=True)[0]
        hold["loss"] = loss                                    # keeps the step's autograd graph alive, like a careless caller
        return loss.mean()
    step1 = GraphedTrainStep(loss_fn, opt, [], warmup=2)      # random coins
    first = [float(step1()) for _ in range(3)]
    dec.gt_as_output = True
    with warnings.catch_warnings():
        warnings.simplefilter("error")                         # torch's "AccumulateGrad node's stream does not match" included
        step2 = GraphedTrainStep(loss_fn, opt, [], warmup=1)
    losses = []
    for i in range(120):
        losses.append(round(float(step2()), 5))
        if i % 40 == 39:
            float(step1())                                     # the older graph is still usable and does not disturb the newer
    assert len(set(losses)) == 1, sorted(set(losses))
    opt.zero_grad()
    eager = float(loss_fn())
    assert abs(eager - losses[0]) <= 2e-5 * max(1.0, abs(eager)), (eager, losses[0])
    assert all(v == v for v in first)
