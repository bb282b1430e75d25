"""The persistent BiLSTM recurrence (csrc/lstm_persist.hip: 64-128 co-resident workgroups handing hidden-state granules to each
other with bounded spins) and the one-pass BatchNorm backward beside ANOTHER live kernel on the same GPU.

What the multi-GPU step adds to the single-GPU one is a collective kernel (RCCL all-reduce of a gradient bucket) running on a side
stream while backward continues.  RCCL cannot put two ranks on one GPU and a 1-rank all-reduce moves nothing, so the hazard --
resident-grid kernels sharing the CUs with a foreign long-running kernel -- is reproduced with what a collective is to the
scheduler: a stream of long kernels that occupy every CU (large GEMMs / copies on a side stream), launched continuously while the
captured CRNN step is replayed 200 times at the 8-GPU strong-scaling shard (32 crops, where the recurrences are 29 % of the step).
Checked after every replay batch: the loss is finite and equal to the undisturbed one (lr = 0), the timeout status words of the
recurrence workspaces stay zero.  Reference of the step: trainer.py:114-130 over structure/model.py:27-36 (apex DDP)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd import _lib  # noqa: E402
from megreader_amd.nn import functional as F  # noqa: E402
from megreader_amd.runtime import GraphedTrainStep  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("crops", [32, 256])
def test_captured_crnn_step_beside_a_busy_side_stream(crops):
    from megreader_amd.backbones import crnn_backbone
    from megreader_amd.decoders import CRNNDecoder
    from megreader_amd.optim import FusedAdam
    from megreader_amd.synthetic import recognition_batch

    mr.set_compute_dtype(torch.bfloat16)
    assert _lib.get_tuning()["lstm_persist"] == 1

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = crnn_backbone()
            self.decoder = CRNNDecoder(in_channels=512, inner_channels=256)

        def forward(self, data, *a, **k):
            return self.decoder(self.backbone(data), *a, **k)

    torch.manual_seed(7)
    model = Model().to(DEV).train()
    opt = FusedAdam(model.parameters(), lr=0.0)
    b = recognition_batch(crops, 32, 128, seed=3)
    img, lab, ln = b['image'].to(DEV), b['label'].to(DEV), b['length'].to(DEV).long()

    def loss_fn(i, l, n):
        loss, _ = model(i, targets=l, lengths=n, train=True)
        return loss.mean()

    F.LSTM_STATUS = status = []
    try:
        graphed = GraphedTrainStep(loss_fn, opt, [img, lab, ln], warmup=2)
    finally:
        F.LSTM_STATUS = None
    assert status, "the step did not use the persistent recurrence"
    for _ in range(5):
        ref = graphed()
    torch.cuda.synchronize()
    ref = float(ref)
    assert ref == ref and abs(ref) < 1e6

    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    big2 = torch.empty_like(big)
    losses = []
    for it in range(20):
        with torch.cuda.stream(side):              # ~2 ms of back-to-back chip-filling kernels per batch of replays
            for _ in range(3):
                torch.mm(a, a)
                big2.copy_(big)
        for _ in range(10):
            losses.append(graphed())
    torch.cuda.synchronize()
    words = [int(t.view(torch.int32).item()) for t in status[-8:]]
    vals = [float(v) for v in losses[-1:]] + [float(losses[0])]
    assert not any(words), ("persistent-recurrence timeout beside the side stream", words)
    for v in vals:
        assert v == v and abs(v - ref) <= 1e-5 * max(1.0, abs(ref)), (v, ref)
