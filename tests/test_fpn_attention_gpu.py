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
