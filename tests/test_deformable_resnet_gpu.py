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


def test_fast_paths_equal_plain_autograd():
    """What the benchmarked step runs and the one-step parity tests do not: from the SECOND training forward on, every conv
    in front of a BatchNorm supplies the batch statistics from its GEMM epilogue, and with a fused optimizer every weight
    gradient -- incl. the 27-channel offset convolutions (Cout < stored channels) and the DCN weights -- is accumulated
    straight into the optimizer's flat buffer.  Same layers, same input: those gradients must equal the plain autograd
    gradients of a first forward without an optimizer (float32; differences = summation order of atomics).
    layer2 only (four deformable bottlenecks, the first strided with a downsample branch): the whole network amplifies the
    1e-7 noise of two summation orders to O(0.3) by itself (two identical plain runs differ that much), layer2 to 7e-6."""
    import copy
    from megreader_amd.optim import FusedSGD
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(1)
    full = deformable_resnet50(pretrained=False)
    perturb_offset_convs(full)                            # non-zero offsets: samples off the bilinear kinks
    plain = full.layer2.to(DEV).train()
    fast = copy.deepcopy(plain)
    x = torch.randn(2, 256, 24, 32, device=DEV)

    def loss_of(model):
        return (model(x).float() ** 2).mean()

    loss_of(plain).backward()
    ref = {k: p.grad.detach().clone() for k, p in plain.named_parameters() if p.grad is not None}
    opt = FusedSGD(fast.parameters(), lr=0.0, momentum=0.0)
    for it in range(2):                                   # 1st: learns the conv -> bn pairs; 2nd: statistics from the epilogue
        opt.zero_grad()
        loss_of(fast).backward()
    producers = [m for m in fast.modules() if getattr(m, "feeds_batch_norm", False)]
    assert len(producers) == 9, len(producers)            # conv1 / conv3 of 4 blocks + the downsample conv (conv2 is the DCN)
    bad, worst = [], (None, 0.0)
    for k, p in fast.named_parameters():
        assert k in ref and p.grad is not None and p.grad.data_ptr() == p._mr_grad_sink.data_ptr(), k      # still the sink
        scale = float(ref[k].abs().max())
        if scale < 1e-9:
            continue
        err = float((p.grad - ref[k]).abs().max()) / scale
        if err > worst[1]:
            worst = (k, err)
        if err > 2e-4:
            bad.append((k, err))
    print("fast paths vs plain autograd: worst gradient difference %.2e of max|g| (%s)" % (worst[1], worst[0]))
    assert not bad, bad[:8]
