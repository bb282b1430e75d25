"""deformable_resnet50 (13 DCNv2 bottlenecks, incl. the stride-2 blocks that index a stride-1 offset map flat --
reference quirk Q10) on HIP.

Golden vectors come from the unmodified reference graph code with the CUDA-only ModulatedDeformConv substituted by the
oracle module (oracle/gen_golden.py deformable_resnet).  At random initialisation this network doubles any
perturbation per block (offsets are computed from features, BatchNorm uses batch statistics), so f32 round-off of
1e-6 after the stem becomes ~1e-2 at layer4: the whole-network comparison is therefore loose, and the strict check is
block-wise (each HIP block on the oracle's own block input, forward and backward).

The block-wise gradient check is exact only while no sample point sits on a kink of the bilinear kernel: the sampling offsets
are themselves computed (conv2_offset) in float32 on both sides, and a point within ~1e-6 of an integer row / column takes
floor() to different sides -- the forward value is continuous there (y still agrees to 1e-6) but d/d(offset) and the set of
input pixels that receive gradient are not.  Seen in round 3 with the block input the oracle produces under
torch.set_num_threads(32) (layer4.0: bn2.bias 25 % off in BOTH DCN code paths, fused and general, while the float32 and
float64 oracles agree to 3e-6); with the default thread count of the box no point is that close.  Tests that change the
thread count must restore it."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.backbones import deformable_resnet50  # noqa: E402
from oracle.dcn import perturb_offset_convs  # noqa: E402
from oracle.res50ppm import _Res50Dilated  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return torch.load(os.path.join(golden_dir, "deformable_resnet50_golden.pt"), weights_only=False)


def _build(golden):
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(golden['weight_seed'])
    model = deformable_resnet50(pretrained=False)
    return model


def test_seeded_init_and_whole_network(golden):
    model = _build(golden)
    assert list(model.state_dict().keys()) == golden['state_keys']
    for k, v in model.state_dict().items():
        s, a = golden['state_checksums'][k]
        assert abs(float(v.double().sum()) - s) <= 1e-6 * max(1.0, a), k   # seeded init == reference
    perturb_offset_convs(model)
    model.to(DEV).train()
    feats = model(golden['x'].to(DEV))
    assert len(feats) == 4
    tols = (1e-4, 1e-3, 1e-2, 5e-2)   # error doubles per block (see module docstring)
    for f, nrm, tol in zip(feats, golden['feature_norms'], tols):
        assert abs(float(f.float().norm()) - nrm) < tol * nrm
    loss = sum(f.float().square().mean() for f in feats)
    assert abs(float(loss) - golden['loss']) < 2e-2 * abs(golden['loss'])
    loss.backward()
    for k, p in model.named_parameters():
        gs = golden['grad_stats'][k]
        assert (p.grad is None) == (gs is None), k
        if gs is not None:
            assert torch.isfinite(p.grad).all(), k


@pytest.mark.parametrize("block", ["layer2.0", "layer2.1", "layer3.0", "layer3.3", "layer4.0", "layer4.2"])
def test_block_parity(golden, block):
    model = _build(golden)
    perturb_offset_convs(model)
    torch.manual_seed(golden['weight_seed'])
    ora = _Res50Dilated(dilate=False, dcn=True)
    perturb_offset_convs(ora)
    ora.train()
    captured = {}
    mod_o = dict(ora.named_modules())[block]
    mod_o.register_forward_pre_hook(lambda m, inp: captured.__setitem__('x', inp[0].detach().clone()))
    ora(golden['x'])
    x = captured['x']
    xo = x.clone().requires_grad_(True)
    yo = mod_o(xo)
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(3))
    yo.backward(g)
    mod_m = dict(model.named_modules())[block].to(DEV).train()
    xm = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ym = mod_m(xm)
    rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))  # noqa: E731
    assert rel(ym, yo) < 1e-4, rel(ym, yo)
    ym.backward(g.to(DEV).contiguous(memory_format=torch.channels_last))
    po = dict(mod_o.named_parameters())
    errs = {k: rel(p.grad, po[k].grad) for k, p in mod_m.named_parameters() if float(po[k].grad.abs().max()) >= 1e-7}
    errs["input"] = rel(xm.grad, xo.grad)
    print("block %s: input-gradient error %.2e, worst parameter-gradient error %.2e (%s)" %
          (block, errs["input"], max(v for k, v in errs.items() if k != "input"),
           max((v, k) for k, v in errs.items() if k != "input")[1]))
    bad = {k: v for k, v in errs.items() if v >= (1e-3 if k == "input" else 2e-3)}
    assert not bad, bad
