"""Parity of the step that is actually TIMED (VERDICT r3 "next round" item 2; reference trainer.py:114-143).

tests/test_published_configs_gpu.py and tests/test_fullsize_parity_gpu.py check the FIRST forward / backward of a freshly
built model with plain autograd.  bench.py (and the graphed drop-in trainer step, megreader_amd/dropin.py) times something
else: step >= 2 under FusedAdam / FusedSGD and a hipGraph replay of it, where
  * every convolution in front of a BatchNorm hands over the batch statistics from its GEMM epilogue (nn/modules.py),
  * every parameter gradient is accumulated straight into the optimizer's flat buffer (gradient sinks, nn/functional.py),
  * the deformable blocks take the packed offset/mask operand, the weight images come from the batched prep launch,
  * the BatchNorm / LSTM scratch comes from the pre-zeroed arena that `zero_grad()` re-zeroes,
  * and -- for the replay -- every launch and every buffer address is the one recorded at capture time.
For each of the four published configurations (BASELINE.json configs[1..4], at the sizes bench.py runs) this file runs
  1. two eager steps under the YAML's optimizer with lr = 0 (weights stay put, so the oracle's gradients remain the truth),
  2. a GraphedTrainStep capture + two replays,
and holds the gradients found in the optimizer's flat buffer after (1) and after (2) to the SAME float64-anchored bars as the
first-step tests (tests/_parity.py: every element of every parameter), plus replay == eager.  The oracle runs are shared with
the first-step tests through tests/_cases.py (one CPU run per configuration and process).

Each configuration's GPU part runs in a FRESH process (the oracle's gradients travel through a temporary file), like
bench.py's own measurement.  Reason: a hipGraph capture can take the whole process down -- the first version of this file kept
the eager step's `loss` alive across the capture and hipStreamEndCapture segfaulted (stale AccumulateGrad nodes created on the
legacy default stream pull that stream into the capture: megreader_amd/runtime.py `_capture`, tools/diag_capture.py,
profiles/r04_diag_capture_crash.txt).  That is fixed, but a crash of this kind must never take the rest of the suite with it.
"""
import os
import subprocess
import time
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
import _cases  # noqa: E402
from _parity import REPORT, grad_report  # noqa: E402
from megreader_amd.runtime import GraphedTrainStep  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


def _timed_step(case, always=(), replay_bar=1e-3):
    mr.set_compute_dtype(torch.float32)
    what = case["what"]
    model = case["build"]()
    batch = _cases.to_device(case["batch"])
    grads32, grads64 = case["grads32"], case["grads64"]
    named = [(k, p) for k, p in model.named_parameters() if k in grads32]
    opt = case["optimizer"](model.parameters())
    loss_fn = case["loss_fn"]
    losses = []
    for it in range(2):             # 1st: pairs conv -> bn, fills the prepared-weight cache; 2nd: the fast paths
        opt.zero_grad()
        loss = loss_fn(model, batch)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        del loss                    # no default-stream autograd graph may be alive when the step is captured (runtime.py)
    producers = sum(1 for m in model.modules() if getattr(m, "feeds_batch_norm", False))
    for k, p in named:
        assert p.grad is not None and p.grad.data_ptr() == p._mr_grad_sink.data_ptr(), k    # still the optimizer's view
    print("%s: loss step 1 %.6f, step 2 %.6f; %d convolutions supply BatchNorm statistics from their epilogue" %
          (what, losses[0], losses[1], producers))
    assert abs(losses[1] - losses[0]) <= 1e-5 * max(1.0, abs(losses[0])), losses      # lr = 0: same weights, same batch
    grad_report(named, grads32, grads64, what + " | step 2, fused optimizer", always=always)
    eager = {k: p.grad.detach().clone() for k, p in named}
    # ---- the same step as ONE captured hipGraph (what bench.py's timed region and the drop-in trainer replay)
    graphed = GraphedTrainStep(lambda: loss_fn(model, batch), opt, [], warmup=1)
    for _ in range(2):
        gl = graphed()
    torch.cuda.synchronize()
    assert abs(float(gl) - losses[1]) <= 1e-5 * max(1.0, abs(losses[1])), (float(gl), losses[1])
    grad_report(named, grads32, grads64, what + " | hipGraph replay", always=always)
    # replay vs eager: the same kernels on the same addresses; what differs is the arrival order of the atomics in the split
    # reductions.  Relative L2 per parameter (one ReLU decision that flips under that noise moves ONE per-pixel term -- large
    # as a single element, invisible in the norm; tests/_parity.py ReluMasks); the largest single element is printed.
    worst_l2, worst_el = (None, 0.0), (None, 0.0)
    for k, p in named:
        if float(grads64[k].abs().max()) < 1e-7 or float(eager[k].abs().max()) < 1e-9:
            continue      # conv biases in front of a BatchNorm: mathematically zero gradient, pure round-off on every side
        d = (p.grad - eager[k]).double()
        l2 = float(d.norm() / eager[k].double().norm())
        el = float(d.abs().max() / eager[k].abs().max())
        if l2 > worst_l2[1]:
            worst_l2 = (k, l2)
        if el > worst_el[1]:
            worst_el = (k, el)
    print("%s: hipGraph replay vs eager step 2: worst relative L2 difference %.2e (%s), worst single element %.2e of max|g| (%s)"
          % (what, worst_l2[1], worst_l2[0], worst_el[1], worst_el[0]))
    REPORT[what + " | replay vs eager"] = {"worst_l2": worst_l2, "worst_elem": worst_el}
    assert worst_l2[1] <= replay_bar, (worst_l2, worst_el)
    del graphed, eager
    _bf16_drift(case, named, losses[1])
    return producers


# bf16 bars per configuration: (relative L2 error, cosine) of a parameter's gradient against the float64 oracle, for parameters
# whose name starts with one of the prefixes, first match wins; "" = everything else.  bf16 is the BENCHMARKED dtype; the
# reference has no bf16 path, so this is drift of another precision (not parity): the bars are ~1.5 x what the replayed step
# measured on the MI355X (profiles/r05_bf16_drift_timed_step.txt) and exist to catch a kernel that drifts further.
# The ResNet-50 backbones sit at the level tools/bf16_storage_sensitivity.py measures for the float64 ORACLE ITSELF when only its
# activations are stored in bf16 (profiles/r05_bf16_storage_sensitivity_of_the_oracle.txt: relative L2 1.3, cosine 0.12 on layers
# 1-3 -- 53 batch-statistics BatchNorms at random initialisation on synthetic crops decorrelate under 8-bit mantissas whoever does
# the arithmetic; the CRNN does not: 0.12 / 0.993), so their bars only fence that level off; the heads are held tight.
BF16_BARS = {
    "CRNN fp32 32x128 N=256": [("backbone.cnn.0", (0.30, 0.97)), ("backbone.cnn.1", (0.22, 0.98)), ("backbone.cnn.", (0.14, 0.992)),
                               ("decoder.", (0.03, 0.9995))],
    "Res50-PPM-2DCTC fp32 32x128 N=256": [("decoder.pred_classify", (0.05, 0.999)), ("decoder.pred_mask", (0.25, 0.975)),
                                          ("backbone.1.conv_last", (0.30, 0.97)), ("backbone.1.ppm", (1.3, 0.4)),
                                          ("backbone.0.layer4", (1.8, 0.0)), ("", (2.4, -0.5))],
    "FPN50-attention fp32 64x256 N=32": [("decoder.decoder.embedding", (0.08, 0.998)), ("decoder.decoder.word_linear", (0.08, 0.998)),
                                         ("decoder.decoder.rnn", (0.2, 0.985)), ("decoder.decoder.out", (0.2, 0.985)),
                                         ("decoder.onehot_embedding", (0.2, 0.985)), ("decoder.decoder.attn", (1.7, 0.1)),
                                         ("", (2.4, -0.5))],
    "DB detector fp32 640x640 N=2": [("decoder.binarize.4", (0.03, 0.9995)), ("decoder.binarize.6", (0.03, 0.9995)),
                                     ("decoder.binarize.1", (0.15, 0.99)), ("decoder.binarize.3", (0.15, 0.99)),
                                     ("decoder.thresh.4", (0.15, 0.99)), ("decoder.thresh.6", (0.15, 0.99)),
                                     ("decoder.thresh.1", (0.9, 0.6)), ("decoder.thresh.3", (0.9, 0.6)),
                                     ("decoder.binarize.0", (0.8, 0.7)), ("decoder.thresh.0", (1.4, 0.3)),
                                     ("decoder.", (1.35, 0.35)), ("", (3.2, -0.6))],
}


def _group_of(name):
    """parameter group of the yardstick rule: backbone stage / block family / head layer"""
    parts = name.split(".")
    if parts[0] == "backbone" and len(parts) > 2:
        return ".".join(parts[:3]) if parts[1] in ("0", "1", "bottom_up", "cnn") or parts[1].startswith("layer") else ".".join(parts[:2])
    return ".".join(parts[:2])


def _bf16_drift(case, named32, loss32):
    """the step that is TIMED runs in bf16.  Same weights, same batch, fused optimizer (lr = 0), two eager
    steps and a hipGraph replay in bf16; every parameter's gradient out of the REPLAY against the float64 oracle."""
    import gc
    gc.collect()
    mr.set_compute_dtype(torch.bfloat16)
    what = case["what"].replace("fp32", "bf16")
    model = case["build"]()
    batch = _cases.to_device(case["batch"])
    grads64 = case["grads64"]
    named = [(k, p) for k, p in model.named_parameters() if k in grads64]
    opt = case["optimizer"](model.parameters())
    loss_fn = case["loss_fn"]
    for _ in range(2):
        opt.zero_grad()
        loss = loss_fn(model, batch)
        loss.backward()
        opt.step()
        l16 = float(loss)
        del loss
    graphed = GraphedTrainStep(lambda: loss_fn(model, batch), opt, [], warmup=1)
    for _ in range(2):
        gl = graphed()
    torch.cuda.synchronize()
    print("%s: loss eager %.6f, replay %.6f (fp32 %.6f)" % (what, l16, float(gl), loss32))
    assert abs(float(gl) - l16) <= 2e-3 * max(1.0, abs(l16)), (float(gl), l16)
    assert abs(l16 - loss32) <= 5e-2 * max(1.0, abs(loss32)), (l16, loss32)
    bars = BF16_BARS.get(case["what"])
    yard = case.get("grads64s")
    rows, bad, groups = [], [], {}
    for k, p in named:
        go, g = grads64[k].double().flatten(), p.grad.double().cpu().flatten()
        if float(go.abs().max()) < 1e-7:
            continue
        l2 = float((g - go).norm() / go.norm())
        cos = float(torch.dot(g, go) / (g.norm() * go.norm() + 1e-300))
        rows.append((k, float(go.norm()), l2, cos))
        if yard is not None and k in yard:
            gs = yard[k].double().flatten()
            grp = groups.setdefault(_group_of(k), {"hip": [], "yard": [], "nh": [], "ny": []})
            grp["hip"].append(l2)
            grp["yard"].append(float((gs - go).norm() / go.norm()))
            grp["nh"].append(float(g.norm() / go.norm()))
            grp["ny"].append(float(gs.norm() / go.norm()))
        if bars is not None:
            for prefix, (l2_bar, cos_bar) in bars:
                if k.startswith(prefix):
                    if cos_bar <= 0.0:
                        break            # a bar that cannot fail: this parameter is held by the yardstick rule below
                    if not (l2 <= l2_bar and cos >= cos_bar):
                        bad.append((k, l2, cos, l2_bar, cos_bar))
                    break
    # The discriminating bar for the deep backbones (VERDICT r5 weak item 1 / task 6e): per parameter group, the HIP bf16 step may
    # be at most 1.5 x as far from exact float64 as the float64 ORACLE ITSELF is once its activations are stored in bf16, and its
    # gradient norms must match that oracle's within 1.5 x -- a kernel that loses a term, doubles one or drifts beyond what bf16
    # storage costs anybody fails; the chaos of 53 batch-statistics BatchNorms at random initialisation (which both share) does not.
    med = lambda v: sorted(v)[len(v) // 2]
    for gname, grp in sorted(groups.items()):
        h, y = med(grp["hip"]), med(grp["yard"])
        nh, ny = med(grp["nh"]), med(grp["ny"])
        print("   group %-34s %3d parameters: l2 vs f64 -- HIP bf16 %.4f, f64 oracle with bf16 storage %.4f (ratio %.2f); "
              "norm ratio %.3f vs %.3f" % (gname, len(grp["hip"]), h, y, h / max(y, 1e-12), nh, ny))
        if not (h <= 1.5 * y + 0.02):
            bad.append(("group " + gname, h, y, "l2 <= 1.5 x yardstick + 0.02"))
        if not (ny / 1.5 <= nh <= ny * 1.5):
            bad.append(("group " + gname, nh, ny, "gradient norm within 1.5 x of the yardstick's"))
    print("%s: per-parameter gradient of the hipGraph replay vs the f64 oracle (relative L2 error, cosine), %d parameters" %
          (what, len(rows)))
    for k, gn, l2, cos in rows:
        print("   %-60s |g| %.3e   l2 %.4f  cos %.5f" % (k, gn, l2, cos))
    worst = sorted(rows, key=lambda r: -r[2])[:5]
    print("BF16-DRIFT %s: worst l2 %s | lowest cosine %s" %
          (what, ", ".join("%s %.3f" % (r[0], r[2]) for r in worst),
           ", ".join("%s %.4f" % (r[0], r[3]) for r in sorted(rows, key=lambda r: r[3])[:5])))
    REPORT[what + " | bf16 replay drift"] = {"worst_l2": worst[0][:3] if worst else None}
    assert not bad, bad[:10]


def _isolated(name, always=(), replay_bar=1e-3):
    """Oracle case in THIS process (memoised), GPU part in a child: `python tests/test_timed_step_gpu.py --child file`."""
    import tempfile
    case = getattr(_cases, name)()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "case.pt")
        t0 = time.time()
        yard = case["yardstick"]()          # float64 oracle with bf16 storage: the yardstick of the bf16 bars (memoised below)
        case["yardstick"] = lambda y=yard: y
        print("oracle %s with bf16 storage (yardstick): %.1f s" % (name, time.time() - t0))
        torch.save({"name": name, "grads32": case["grads32"], "grads64": case["grads64"], "grads64s": yard,
                    "always": tuple(always), "replay_bar": replay_bar}, path)
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([repo, os.path.join(repo, "tests"), os.environ.get("PYTHONPATH", "")]))
        out = subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "--child", path], cwd=repo, env=env,
                             capture_output=True, text=True, timeout=900)
    text = out.stdout + out.stderr
    print("\n".join(ln for ln in out.stdout.splitlines() if not ln.startswith("   ")))
    dump = os.environ.get("MEGREADER_TIMED_STEP_DUMP")
    if dump:      # the per-parameter tables (lines starting with three blanks) for profiles/
        with open(dump, "a") as f:
            f.write(out.stdout)
    assert out.returncode == 0, "child exited with %d\n%s" % (out.returncode, text[-3000:])
    res = [ln for ln in out.stdout.splitlines() if ln.startswith("TIMED-STEP-OK ")]
    assert res, text[-2000:]
    return int(res[-1].split()[1])


def _child(path):
    """The GPU part: models are rebuilt from the case's seed (weights = the oracle's initial state), gradients come from the file."""
    blob = torch.load(path, weights_only=False)
    name = blob["name"]
    case = dict(getattr(_cases, name + "_hip")())
    case["grads32"], case["grads64"] = blob["grads32"], blob["grads64"]
    case["grads64s"] = blob.get("grads64s")
    producers = _timed_step(case, always=blob["always"], replay_bar=blob["replay_bar"])
    print("TIMED-STEP-OK %d" % producers)


def test_crnn_timed_step():
    assert _isolated("crnn_n256") == 3                    # cnn.2 / cnn.4 / cnn.6: conv -> BatchNorm (no ReLU in between)


def test_res50ppm_timed_step():
    assert _isolated("res50ppm_n256") >= 50               # 53 batch-statistics BatchNorms, all behind a convolution


def test_fpn_attention_timed_step():
    _isolated("fpn_attention_n32", always=("decoder.decoder.",))


def test_db_detector_timed_step():
    # whole deformable network at default initialisation.  Round 4: the small layers' tap-split forward added its partial sums
    # with f32 atomics, eager and replay differed by round-off in the FORWARD pass, which this network amplifies (module docstring
    # of tests/test_deformable_resnet_gpu.py), and this bar was 5e-2.  Round 5: per-tap-group slabs summed in tap order
    # (csrc/dcn_fused.hip: dcn_finish_kernel) -- the forward pass is the same bits every run, the default bar applies again
    _isolated("db_n2", always=("conv2_offset.weight", "layer4.2.conv2."))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        _child(sys.argv[2])
