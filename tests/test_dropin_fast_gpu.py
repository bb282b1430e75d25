"""The FAST drop-in: `dropin.install(fused_optimizer=True, graph_step=True)` gives the reference's unchanged
entry point (trainer.py:114-143 + training/optimizer_scheduler.py:17-22) the step bench.py measures -- torch.optim.Adam
resolves to FusedAdam and Trainer.train_step replays one captured hipGraph.

The reference tree does not exist on the GPU box, so the trainer / model wrapper here are line-for-line facsimiles of
trainer.py:114-130 (`train_step`) and structure/model.py:160-181 (`SequenceRecognitionModel.forward`: batch dict of CPU
tensors, `.to(device)` inside forward); tests/test_b6_dropin_cpu.py runs the real files in the build container."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd import dropin  # noqa: E402
from megreader_amd.backbones import crnn_backbone  # noqa: E402
from megreader_amd.decoders import CRNNDecoder  # noqa: E402
from megreader_amd.synthetic import recognition_batch  # noqa: E402

DEV = torch.device("cuda")


class BasicModel(torch.nn.Module):                       # structure/model.py:16-24
    def __init__(self):
        super().__init__()
        self.backbone = crnn_backbone()
        self.decoder = CRNNDecoder(in_channels=512, inner_channels=256)

    def forward(self, data, *args, **kwargs):
        return self.decoder(self.backbone(data), *args, **kwargs)


class SequenceRecognitionModel(torch.nn.Module):         # structure/model.py:160-181
    def __init__(self, device):
        super().__init__()
        self.model = BasicModel()
        self.device = device
        self.to(self.device)

    def forward(self, batch, training=True):
        images = batch['image'].to(self.device)
        if self.training:
            labels = batch['label'].to(self.device)
            lengths = batch['length'].to(self.device).type(torch.long)
            loss, pred = self.model(images, targets=labels, lengths=lengths, train=True)
            return loss, pred
        return self.model(images, train=False)


class MiniTrainer(object):                                # trainer.py:114-130 without the logging block
    device = DEV

    def train_step(self, model, optimizer, batch, epoch, step, **kwards):
        optimizer.zero_grad()
        results = model.forward(batch, training=True)
        if len(results) == 2:
            l, pred = results
        else:
            l = results
        loss = l.mean()
        loss.backward()
        optimizer.step()
        return loss


@pytest.fixture(autouse=True)
def _restore():
    adam, sgd = torch.optim.Adam, torch.optim.SGD
    yield
    torch.optim.Adam, torch.optim.SGD = adam, sgd
    mr.set_compute_dtype(torch.bfloat16)


def test_optimizer_aliasing():
    from megreader_amd.optim import FusedAdam, FusedSGD
    dropin.fuse_optimizers()
    dropin.fuse_optimizers()                               # idempotent
    w = torch.nn.Parameter(torch.randn(8, 8, device=DEV))
    assert isinstance(getattr(torch.optim, 'Adam')([w], lr=1e-3), FusedAdam)           # optimizer_scheduler.py:18
    assert isinstance(torch.optim.SGD([w], lr=0.007, momentum=0.9, weight_decay=1e-4), FusedSGD)
    assert not isinstance(torch.optim.Adam([w], lr=1e-3, amsgrad=True), FusedAdam)       # unsupported option: torch's own
    wc = torch.nn.Parameter(torch.randn(4))
    assert not isinstance(torch.optim.Adam([wc], lr=1e-3), FusedAdam)                    # CPU parameters: torch's own


def _run(accelerated, batches, steps, dtype):
    mr.set_compute_dtype(dtype)
    torch.manual_seed(0)
    model = SequenceRecognitionModel(DEV).train()
    opt = getattr(torch.optim, 'Adam')(model.parameters(), lr=1e-3)

    class T(MiniTrainer):
        pass
    wrapper = dropin.accelerate_trainer(T, eager_steps=3) if accelerated else None
    tr = T()
    # the last 10 steps are timed with events on the main stream and WITHOUT a host synchronisation in front of them: a
    # sync would let the GPU run dry, and the first timed step would then pay its 25 MB host-to-device copy un-overlapped
    # (0.5 - 2 ms depending on the box's PCIe link) -- in steady state the host runs ahead and the copy of step s + 1 rides
    # beside the replay of step s
    losses = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for s in range(steps):
        if s == steps - 10:
            e0.record()
        losses.append(tr.train_step(model, opt, batches[s % len(batches)], epoch=0, step=s).detach())
    e1.record()
    torch.cuda.synchronize()
    dt = 1e-3 * e0.elapsed_time(e1) / 10
    return [float(l) for l in losses], dt, wrapper


def _pinned_batches(n, count):
    # CPU tensors in pinned memory, like the reference DataLoader's (data/data_loader.py:46 pin_memory=True)
    return [{k: v.pin_memory() for k, v in recognition_batch(n, 32, 128, seed=100 + i).items()} for i in range(count)]


def test_graphed_train_step_equals_eager_trajectory():
    """Same kernels in the same order: the loss trajectory of 16 Adam steps (3 eager, then replays of the captured step on
    changing batches) equals the all-eager trajectory.  float32 compute, so that the only difference left is the summation
    order of the atomically reduced weight gradients."""
    dropin.fuse_optimizers()
    batches = _pinned_batches(32, 4)
    eager, _, _ = _run(False, batches, 16, torch.float32)
    fast, _, wrapper = _run(True, batches, 16, torch.float32)
    assert wrapper.state is not None and not wrapper.disabled, "the step was never captured"
    print("drop-in trajectory: eager", ["%.4f" % v for v in eager])
    print("drop-in trajectory: graph", ["%.4f" % v for v in fast])
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(eager[:3], fast[:3]))   # both eager: equal up to atomics order
    for a, b in zip(eager, fast):
        assert abs(a - b) <= 5e-3 * max(1.0, abs(a)), (eager, fast)
    assert fast[-1] < fast[0]


def _measure_speed():
    """eager / graphed drop-in / bench.py-style replay step times (seconds) at bf16, N = 256 (BASELINE.json configs[1])."""
    dropin.fuse_optimizers()
    batches = _pinned_batches(256, 4)
    _, t_eager, _ = _run(False, batches, 24, torch.bfloat16)
    _, t_fast, wrapper = _run(True, batches, 24, torch.bfloat16)
    captured = wrapper.state is not None and not wrapper.disabled
    from megreader_amd.runtime import GraphedTrainStep
    torch.manual_seed(0)
    model = SequenceRecognitionModel(DEV).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    dbatch = {k: v.to(DEV) for k, v in batches[0].items()}
    g = GraphedTrainStep(lambda: model.forward(dbatch, training=True)[0].mean(), opt, [], warmup=3)
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g()
    torch.cuda.synchronize()
    t_bench = (time.perf_counter() - t0) / 10
    return {"t_eager": t_eager, "t_fast": t_fast, "t_bench": t_bench, "captured": captured}


def test_graphed_train_step_is_as_fast_as_bench():
    """bf16, N = 256 (BASELINE.json configs[1]): the drop-in step (pinned host batch in, H2D overlapped with the previous
    replay) against bench.py's own GraphedTrainStep on a device-resident batch, and against the eager trainer step.
    Measured in a FRESH process, like bench.py itself: late in a long pytest session torch hands out the copy stream from a
    pool that earlier tests have cycled through, HIP maps it onto the hardware queue the main stream already uses, and the
    25 MB host-to-device copy then serialises with the replay (+0.25 ms = 12.6 MB at PCIe speed; three suite runs measured
    3.12-3.15 vs 2.86-2.91 ms, the same test alone 2.875 vs 2.833 ms)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--measure"], cwd=repo, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    t_eager, t_fast, t_bench = r["t_eager"], r["t_fast"], r["t_bench"]
    assert r["captured"], "the step was never captured"
    print("drop-in train_step: eager %.3f ms, graphed %.3f ms (pinned 25 MB host batch per step); bench.py-style replay on a "
          "resident batch %.3f ms" % (1e3 * t_eager, 1e3 * t_fast, 1e3 * t_bench))
    # the H2D copy of the next batch overlaps the replay of the current one: the drop-in step costs what bench.py measures
    # + a 25 MB device-to-device copy and the loss clone (measured 2.875 vs 2.833 ms = +1.5 %; VERDICT r2 item 7 asks for 5 %)
    assert t_fast < 1.05 * t_bench + 0.05e-3, (t_fast, t_bench)
    assert t_fast < t_eager, (t_fast, t_eager)     # eager with a fused optimizer and pinned batches: 3.2 - 3.4 ms


if __name__ == "__main__":
    import json
    import sys
    if "--measure" in sys.argv:
        print(json.dumps(_measure_speed()))
