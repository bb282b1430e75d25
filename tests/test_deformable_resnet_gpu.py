"""deformable_resnet50 (13 DCNv2 bottlenecks, incl. the stride-2 blocks that index a stride-1 offset map flat --
reference quirk Q10) on HIP vs golden vectors from the unmodified reference graph code with the CUDA-only
ModulatedDeformConv substituted by the oracle module (oracle/gen_golden.py deformable_resnet)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.backbones import deformable_resnet50  # noqa: E402
from oracle.dcn import perturb_offset_convs  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


def test_fp32_parity_vs_reference_golden(golden_dir):
    g = torch.load(os.path.join(golden_dir, "deformable_resnet50_golden.pt"), weights_only=False)
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(g['weight_seed'])
    model = deformable_resnet50(pretrained=False)
    assert list(model.state_dict().keys()) == g['state_keys']
    for k, v in model.state_dict().items():
        s, a = g['state_checksums'][k]
        assert abs(float(v.double().sum()) - s) <= 1e-6 * max(1.0, a), k   # seeded init == reference
    perturb_offset_convs(model)
    model.to(DEV).train()
    feats = model(g['x'].to(DEV))
    assert len(feats) == 4
    for f, nrm in zip(feats, g['feature_norms']):
        assert abs(float(f.float().norm()) - nrm) < 2e-3 * nrm
    for f, ref in zip(feats[2:], g['features']):     # x4, x5 stored in full
        assert f.shape == ref.shape
        assert float((f.float().cpu() - ref).abs().max()) < 5e-3 * max(1.0, float(ref.abs().max()))
    loss = sum(f.float().square().mean() for f in feats)
    assert abs(float(loss) - g['loss']) < 1e-3 * abs(g['loss'])
    loss.backward()
    worst = 0.0
    for k, p in model.named_parameters():
        gs = g['grad_stats'][k]
        if gs is None:
            assert p.grad is None, k
            continue
        norm, _ = gs
        if norm < 1e-6:
            continue
        rel = abs(float(p.grad.double().norm()) - norm) / norm
        worst = max(worst, rel)
        assert rel < 5e-2, (k, rel, norm)
    print("worst relative grad-norm error:", worst)
