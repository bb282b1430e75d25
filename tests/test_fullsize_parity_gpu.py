"""Parity at the BENCHMARKED configurations: the HIP path at the sizes bench.py times -- where the
library selects different kernels than at the tiny golden batch (96x128 / 8-wave 256x256 NT tiles, XCD-aware maps with
>1000 tiles, split-P wgrad with row tables, the persistent BiLSTM recurrence) -- against the CPU oracle on the same
seeded weights and batch.  Full tensors: loss, every log-probability, EVERY element of EVERY parameter gradient,
greedy decode (1-D and 2-D rule) with the minimum top-1/top-2 margin reported.

Bars (BASELINE.json north_star: loss/logits within 1e-4 fp32, decode bit-exact):
  fp32  loss |d| <= 1e-4, log-probs max|d| <= 1e-4 against the f32 oracle (= what the reference computes).
        Gradients: a weight gradient of the first layers is a 1M-term f32 sum with heavy cancellation, so the CPU's own
        f32 result is only good to ~1e-3 of max|g| there; the ground truth is therefore the oracle run in FLOAT64 and the
        bar is "HIP-f32 is as close to it as the reference's own f32 arithmetic": for every parameter
        max|g_hip - g_64| <= max(4 * max|g_cpu32 - g_64|, 2e-3 * max|g_64|)   (element-wise; round 4: the flat 1e-2 floor
        became 2e-3 plus a counted census of at most max(3, min(16, 2 %)) isolated elements up to 1e-2, tests/_parity.py) and
        ||g_hip - g_64||_2 <= max(4 * ||g_cpu32 - g_64||_2, 2e-3 * ||g_64||_2).
        The census cap is the size of ONE discontinuity event: ReLU and max-pool are not continuous, so a single
        pre-activation whose f32 value rounds across zero (or a near-tie in a pooling window) moves a whole per-pixel
        gradient term -- 0.5 % of an element of the cnn.3 / cnn.5 bias gradients at this batch; the CPU f32 oracle shows
        the same events (cnn.3 bias 5.9e-4, conv weights 1e-3..5e-3), just not always in the same tensors.  An indexing
        or layout bug moves elements by O(max|g|) and is far outside both bars.  Both errors are printed per parameter.
  bf16  (the benchmarked dtype; the reference has no bf16 path, so this is drift of a different precision, not parity)
        loss within 2e-2; per-parameter relative L2 error and cosine against the f64 gradients are printed as a table
        and bounded per depth (see the test).
"""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.backbones import crnn_backbone, resnet50dilated_ppm  # noqa: E402
from megreader_amd.decoders import CRNNDecoder, CTCDecoder2D  # noqa: E402
from oracle.crnn import CRNNOracle, synthetic_batch  # noqa: E402
from oracle.decode import greedy_decode, greedy_decode_2d  # noqa: E402
from oracle.res50ppm import Res50PPM2DCTCOracle, synthetic_batch_2d  # noqa: E402
from _parity import REPORT  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


class CRNNModel(torch.nn.Module):  # reference structure/model.py:16-24
    def __init__(self):
        super().__init__()
        self.backbone = crnn_backbone()
        self.decoder = CRNNDecoder(in_channels=512, inner_channels=256, need_reduce=False)

    def forward(self, data, *args, **kwargs):
        return self.decoder(self.backbone(data), *args, **kwargs)


class Res50Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = resnet50dilated_ppm()
        self.decoder = CTCDecoder2D(in_channels=256)

    def forward(self, data, *args, **kwargs):
        return self.decoder(self.backbone(data), *args, **kwargs)




def _f64_grads(ora32, batch):
    """The same model in float64 (weights converted exactly): ground truth for the gradient comparison."""
    import copy
    ora64 = copy.deepcopy(ora32).double().train()
    ora64.zero_grad()
    loss, _ = ora64(batch['image'].double(), targets=batch['label'], lengths=batch['length'].long(), train=True)
    loss.mean().backward()
    return {k: p.grad.detach().clone() for k, p in ora64.named_parameters() if p.grad is not None}


def _crnn_oracle_run():
    """One oracle training forward/backward (f32 = the reference's arithmetic, and f64 = ground truth for gradients)
    + eval at N=256, 32x128 (about 15 s of CPU), memoised in tests/_cases.py and shared with tests/test_timed_step_gpu.py."""
    import _cases
    c = _cases.crnn_n256()
    return (c["state0"], c["state1"], c["batch"], c["out32"]["loss"], c["out32"]["logp"], (c["grads32"], c["grads64"]),
            c["out32"]["eval"])


def _grad_report(named_params, grads_pair, what):
    """Element-wise gradient comparison of every parameter against the float64 oracle, next to the f32 oracle's own error against
    it: tests/_parity.py grad_report (2e-3 floor + census of isolated discontinuity events, round 4)."""
    from _parity import grad_report
    grads32, grads64 = grads_pair
    grad_report([(k, p) for k, p in named_params if k in grads64], grads32, grads64, what)


def test_crnn_fp32_full_batch_elementwise():
    state0, state1, batch, loss_o, logp_o, grads_o, ev_o = _crnn_oracle_run()
    mr.set_compute_dtype(torch.float32)
    model = CRNNModel()
    model.load_state_dict(state0)
    model.to(DEV).train()
    img, lab, ln = batch['image'].to(DEV), batch['label'].to(DEV), batch['length'].to(DEV).long()
    loss, pred = model(img, targets=lab, lengths=ln, train=True)
    assert abs(float(loss) - loss_o) < 1e-4, (float(loss), loss_o)
    err = float((pred.cpu() - logp_o).abs().max())
    print("CRNN fp32 N=256: loss |d| %.2e, log-prob max|d| %.2e" % (abs(float(loss) - loss_o), err))
    assert err < 1e-4
    loss.mean().backward()
    _grad_report(model.named_parameters(), grads_o, "CRNN fp32 N=256")
    for k, v in model.state_dict().items():
        if 'running' in k:
            assert float((v.cpu() - state1[k]).abs().max()) < 1e-4 * max(1.0, float(state1[k].abs().max())), k
    # ---- eval + greedy decode (structure/representers/ctc_representer.py:20-34) on the full batch
    model.eval()
    with torch.no_grad():
        ev = model(img, train=False)
    perr = float((ev.cpu().double() - ev_o.double()).abs().max())
    top2 = ev_o.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])[:, 0, :]                # [N, T]
    print("CRNN fp32 eval: prob max|d| %.2e; min top-1/top-2 margin %.3e" % (perr, float(margin.min())))
    assert perr < 1e-4
    am, am_o = ev.cpu().argmax(dim=1)[:, 0, :], ev_o.argmax(dim=1)[:, 0, :]
    flipped = am != am_o
    # an arg-max may only differ where the oracle's own margin is inside the f32 parity error
    assert bool((margin[flipped] <= 2 * perr).all()), "arg-max differs at a position with a safe margin"
    if not bool(flipped.any()):
        assert np.array_equal(greedy_decode(ev.cpu().numpy()), greedy_decode(ev_o.numpy()))
    n_flip, n_unsafe = int(flipped.sum()), int((margin <= 2 * perr).sum())
    print("CRNN greedy decode: %d of %d positions inside the error margin, %d arg-max flips, decode %s" %
          (n_unsafe, margin.numel(), n_flip, "bit-exact" if n_flip == 0 else "margin-limited"))
    REPORT["CRNN fp32 greedy decode"] = {"positions": margin.numel(), "inside_margin": n_unsafe, "flips": n_flip}
    # the escape hatch is bounded: at most one position in 1000 may sit inside the error margin at all (an untrained net's
    # near-ties), so a systematic arg-max difference cannot hide behind it
    assert n_unsafe <= max(1, margin.numel() // 1000) and n_flip <= n_unsafe


def test_crnn_bf16_full_batch_elementwise():
    state0, _, batch, loss_o, logp_o, grads_o, _ = _crnn_oracle_run()
    mr.set_compute_dtype(torch.bfloat16)
    model = CRNNModel()
    model.load_state_dict(state0)
    model.to(DEV).train()
    img, lab, ln = batch['image'].to(DEV), batch['label'].to(DEV), batch['length'].to(DEV).long()
    loss, pred = model(img, targets=lab, lengths=ln, train=True)
    loss.mean().backward()
    print("CRNN bf16 N=256: loss drift %.3e, log-prob max|d| %.3e" %
          (abs(float(loss) - loss_o), float((pred.cpu() - logp_o).abs().max())))
    assert abs(float(loss) - loss_o) < 2e-2
    grads64 = grads_o[1]
    print("CRNN bf16 N=256: per-parameter gradient vs the f64 oracle (relative L2 error, cosine)")
    bad = []
    for k, p in model.named_parameters():
        go, g = grads64[k].double().flatten(), p.grad.double().cpu().flatten()
        if float(go.abs().max()) < 1e-7:
            continue
        l2 = float((g - go).norm() / go.norm())
        cos = float(torch.dot(g, go) / (g.norm() * go.norm()))
        print("   %-52s |g| %.3e   l2 %.4f  cos %.5f" % (k, float(go.norm()), l2, cos))
        # bf16 rounding noise of seven conv layers' activations and gradients accumulates towards the input: the
        # decoder and the upper conv layers must be tight, the first layers are bounded loosely (and reported)
        deep = k.startswith("backbone.cnn.0") or k.startswith("backbone.cnn.1") or k.startswith("backbone.cnn.2")
        if l2 > (0.6 if deep else 0.15) or cos < (0.85 if deep else 0.985):
            bad.append((k, l2, cos))
    assert not bad, bad


@pytest.mark.parametrize("height,width,n", [(32, 128, 32), (64, 256, 32)])
def test_res50ppm_2dctc_fp32_elementwise(height, width, n):
    """ResNet50-dilated-PPM + 2D-CTC (BASELINE configs[2]) at the north-star crop size and at the YAML-native 64x256
    (community-base.yaml:33-35), batch 32: loss per sample, log-probs, every gradient element, 2-D greedy decode."""
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(99)
    ora = Res50PPM2DCTCOracle(dropout=0.0)
    model = Res50Model()
    model.load_state_dict(ora.state_dict(), strict=True)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    model.to(DEV).train()
    # labels short enough for W/8 time steps: L + repeats <= T
    batch = synthetic_batch_2d(n, height, width, seed=5, max_len=3 if width == 128 else 8)
    t0 = time.time()
    grads64 = _f64_grads(ora, batch)
    ora.train()
    loss_o, pred_o = ora(batch['image'], targets=batch['label'], lengths=batch['length'].long(), train=True)
    loss_o.mean().backward()
    print("oracle Res50-PPM-2DCTC %dx%d N=%d fwd+bwd (f32 and f64): %.1f s" % (height, width, n, time.time() - t0))
    img, lab, ln = batch['image'].to(DEV), batch['label'].to(DEV), batch['length'].to(DEV).long()
    loss, pred = model(img, targets=lab, lengths=ln, train=True)
    lerr = float(((loss.cpu() - loss_o).abs() / loss_o.abs().clamp_min(1.0)).max())
    finite = torch.isfinite(pred_o) & (pred_o > -80)
    perr = float((pred.cpu() - pred_o)[finite].abs().max())
    print("Res50-PPM-2DCTC fp32 %dx%d: loss rel |d| %.2e, log-prob max|d| %.2e" % (height, width, lerr, perr))
    assert lerr < 1e-4
    assert perr < 1e-3
    loss.mean().backward()
    grads_o = {k: p.grad for k, p in ora.named_parameters() if p.grad is not None}
    named = [(k, p) for k, p in model.named_parameters() if k in grads_o]
    for k, p in model.named_parameters():
        if k not in grads_o:
            assert p.grad is None, k      # unused parameters (cbr_deepsup) receive no gradient on either side
    _grad_report(named, (grads_o, grads64), "Res50-PPM-2DCTC fp32 %dx%d" % (height, width))
    # ---- eval + the 2-D decode rule (structure/representers/ctc_representer2d.py:27-51)
    ora.eval()
    model.eval()
    with torch.no_grad():
        cls_o, mask_o = ora(batch['image'], train=False)
        cls, mask = model(img, train=False)
    cerr = float((cls.cpu() - cls_o).abs().max())
    merr = float((mask.cpu() - mask_o).abs().max())
    heat_o = cls_o * mask_o
    heat = cls.cpu().float() * mask.cpu().float()
    herr = float((heat - heat_o).abs().max())
    # margins of the two arg-max decisions of the rule: row pick (over max_c) and class pick at the chosen row
    rowscore = heat_o.max(dim=1).values                      # [N, H, W]
    r2 = rowscore.topk(2, dim=1).values
    row_margin = r2[:, 0] - r2[:, 1]                         # [N, W]
    hstar = rowscore.argmax(dim=1)                           # [N, W]
    sel = heat_o.gather(2, hstar[:, None, None, :].expand(-1, heat_o.shape[1], 1, -1))[:, :, 0, :]   # [N, C, W]
    c2 = sel.topk(2, dim=1).values
    cls_margin = c2[:, 0] - c2[:, 1]
    print("Res50-PPM-2DCTC eval: classify max|d| %.2e, mask max|d| %.2e, heatmap max|d| %.2e; min row margin %.3e, "
          "min class margin %.3e" % (cerr, merr, herr, float(row_margin.min()), float(cls_margin.min())))
    assert cerr < 1e-3 and merr < 1e-3
    dec_o = greedy_decode_2d(cls_o.numpy(), mask_o.numpy())
    dec = greedy_decode_2d(cls.cpu().float().numpy(), mask.cpu().float().numpy())
    safe = bool((row_margin > 2 * herr).all()) and bool((cls_margin > 2 * herr).all())
    if safe:
        assert np.array_equal(dec, dec_o), "2-D greedy decode differs although every arg-max margin is safe"
    else:
        # rows whose every column has safe margins must still decode identically
        ok_rows = ((row_margin > 2 * herr) & (cls_margin > 2 * herr)).all(dim=1).numpy()
        assert np.array_equal(dec[ok_rows], dec_o[ok_rows])
    n_safe = int(((row_margin > 2 * herr) & (cls_margin > 2 * herr)).all(dim=1).sum())
    n_diff = int((dec != dec_o).any(axis=tuple(range(1, dec.ndim))).sum()) if dec.shape == dec_o.shape else n
    print("2-D greedy decode: %s (%d of %d samples with every margin safe, %d samples decode differently)" %
          ("bit-exact" if n_diff == 0 else "margin-limited", n_safe, n, n_diff))
    REPORT["Res50-PPM fp32 2-D greedy decode"] = {"samples": n, "all_margins_safe": n_safe, "differ": n_diff}
    # bounded escape hatch: only samples with an unsafe margin may differ, and those are at most 2 % of the batch
    assert n_diff <= n - n_safe and n - n_safe <= max(1, n // 50)
