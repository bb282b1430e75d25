"""DB head (decoders.SegDetector mirror: HIP convs / BN / nearest upsampling, ConvTranspose2d as a GEMM) against the
oracle restatement (oracle/seg_detector.py, bit-identical to the unmodified reference on CPU -- tests/
test_oracle_models.py): forward maps, loss and every parameter / input gradient, fp32; plus one training step of the
whole DB detector (deformable ResNet-50 + head + loss, BASELINE.json configs[4]) in bf16."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.decoders import L1BalanceCELoss, SegDetector  # noqa: E402
from megreader_amd.synthetic import detection_batch  # noqa: E402
from oracle.seg_detector import SegDetectorOracle, l1_balance_ce_loss  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_seg_detector_fp32_vs_oracle():
    mr.set_compute_dtype(torch.float32)
    chans = [16, 32, 64, 128]
    torch.manual_seed(3)
    ora = SegDetectorOracle(in_channels=chans, inner_channels=64, k=50, adaptive=True).double().train()
    model = SegDetector(in_channels=chans, inner_channels=64, k=50, adaptive=True)
    model.load_state_dict({k: v.float() for k, v in ora.state_dict().items()}, strict=True)
    model.to(DEV).train()
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(2, c, 64 // s, 64 // s, generator=g) for c, s in zip(chans, (1, 2, 4, 8))]
    batch = detection_batch(2, 256, seed=1, boxes=3)
    fo = [f.double().requires_grad_(True) for f in feats]
    po = ora(fo)
    lo = l1_balance_ce_loss(po, {k: v.double() for k, v in batch.items()})
    lo.backward()
    fd = [f.to(DEV).requires_grad_(True) for f in feats]
    pd = model(fd)
    for k in po:
        assert pd[k].shape == po[k].shape and _rel(pd[k], po[k]) < 2e-4, k
    ld, _ = L1BalanceCELoss()(pd, {k: v.to(DEV) for k, v in batch.items()})
    assert abs(float(ld) - float(lo)) < 1e-4 * max(1.0, abs(float(lo)))
    ld.backward()
    for a, b in zip(fd, fo):
        assert _rel(a.grad, b.grad) < 2e-3
    op = dict(ora.named_parameters())
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert _rel(p.grad, op[k].grad) < 5e-3, (k, _rel(p.grad, op[k].grad))


def test_db_detector_training_step_bf16():
    from megreader_amd.backbones import deformable_resnet50
    from megreader_amd.optim import FusedSGD
    mr.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    backbone = deformable_resnet50(pretrained=False).to(DEV).train()
    head = SegDetector(in_channels=[256, 512, 1024, 2048], adaptive=True, k=50).to(DEV).train()
    crit = L1BalanceCELoss()
    params = list(backbone.parameters()) + list(head.parameters())
    opt = FusedSGD(params, lr=0.007, momentum=0.9, weight_decay=1e-4)
    batch = {k: v.to(DEV) for k, v in detection_batch(2, 256, seed=0, boxes=4).items()}
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss, metrics = crit(head(backbone(batch['image'])), batch)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l and abs(l) < 1e3 for l in losses), losses
    assert set(metrics) == {'bce_loss', 'thresh_loss', 'l1_loss'}
