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


def test_graphed_train_step_equals_eager_and_is_as_fast_as_bench():
    mr.set_compute_dtype(torch.bfloat16)
    dropin.fuse_optimizers()
    N, steps = 256, 24
    batches = [recognition_batch(N, 32, 128, seed=100 + i) for i in range(4)]      # CPU tensors, like the DataLoader's

    def run(accelerated):
        torch.manual_seed(0)
        model = SequenceRecognitionModel(DEV).train()
        opt = getattr(torch.optim, 'Adam')(model.parameters(), lr=1e-3)

        class T(MiniTrainer):
            pass
        wrapper = dropin.accelerate_trainer(T, eager_steps=3) if accelerated else None
        tr = T()
        losses, t_tail = [], None
        for s in range(steps):
            if s == steps - 10:
                torch.cuda.synchronize()
                t_tail = time.perf_counter()
            losses.append(tr.train_step(model, opt, batches[s % 4], epoch=0, step=s))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t_tail) / 10
        return [float(l) for l in losses], dt, wrapper

    eager, t_eager, _ = run(False)
    fast, t_fast, wrapper = run(True)
    assert wrapper.state is not None and not wrapper.disabled, "the step was never captured"
    print("drop-in train_step: eager %.3f ms / step, graphed %.3f ms / step (incl. the H2D copy of a 25 MB batch)" %
          (1e3 * t_eager, 1e3 * t_fast))
    # same kernels, same order; the split-P weight-gradient reductions use atomics, so trajectories agree to rounding
    for a, b in zip(eager, fast):
        assert abs(a - b) <= 2e-2 * max(1.0, abs(a)), (eager, fast)
    assert abs(eager[0] - fast[0]) < 1e-6 and fast[-1] < fast[0]
    # bench.py's own step for comparison: GraphedTrainStep on a device-resident batch
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
    print("bench.py-style graph replay on a resident batch: %.3f ms / step" % (1e3 * t_bench))
    # the drop-in pays the H2D copy of the batch (pageable host memory: ~25 MB per step) on top of the replay
    assert t_fast < 1.35 * t_bench + 2.5e-3, (t_fast, t_bench)
    assert t_fast < 0.8 * t_eager, (t_fast, t_eager)
