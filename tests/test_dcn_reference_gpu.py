"""The HIP deformable-convolution / PS-RoI-pooling kernels against the REFERENCE's own extension.

Two legs:
  * fixture (always runs on a GPU box): tests/golden/dcn_reference_ext.npz holds inputs and outputs of the reference's
    `deform_conv_cuda` / `deform_pool_cuda` modules (compiled for gfx950 from /root/reference/assets/ops/dcn/src by
    oracle/build_ref_ext.sh, run on an MI355X by oracle/gen_golden_dcn.py); the same calls go to
    megreader_amd.assets.ops.dcn.{deform_conv_cuda, deform_pool_cuda} in float32;
  * live (when oracle/_ref travelled with the snapshot): both extension modules run side by side on the 13 real layer shapes
    of deformable ResNet-50 at the DB detector's 640 x 640 input, batch 2 (experiments/seg_detector/*.yaml).
Bars (float32 both sides, different summation orders, float atomics in the reference's backward): forward 5e-5 of max|y|,
gradients 2e-4 of the tensor's max."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.assets.ops.dcn import deform_conv_cuda as hip_conv, deform_pool_cuda as hip_pool  # noqa: E402
from oracle.deform_pool import random_case  # noqa: E402
from oracle.gen_golden_dcn import (DCN1_CASES, DCN2_CASES, POOL_CASES, load_reference_extension, run_dcn1, run_dcn2,  # noqa: E402
                                   run_pool)

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "dcn_reference_ext.npz")
needs_fixture = pytest.mark.skipif(not os.path.exists(FIXTURE), reason="fixture not generated yet (oracle/gen_golden_dcn.py)")


@pytest.fixture(autouse=True)
def _fp32():
    mr.set_compute_dtype(torch.float32)
    yield
    mr.set_compute_dtype(torch.bfloat16)


def _case(prefix):
    z = np.load(FIXTURE)
    return {k[len(prefix) + 1:]: z[k] for k in z.files if k.startswith(prefix + "/")}


def _rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@needs_fixture
@pytest.mark.parametrize("name", [c[0] for c in DCN2_CASES] + ["kink"])
def test_dcn2_hip_extension_equals_reference_extension(name):
    c = _case("dcn2/" + name)
    stride, pad, dil = (int(v) for v in c["geom"])
    b = torch.from_numpy(c["bias"]) if "bias" in c else None
    got = run_dcn2(hip_conv, torch.from_numpy(c["x"]), torch.from_numpy(c["offset_mask_map"]), torch.from_numpy(c["weight"]),
                   b, torch.from_numpy(c["grad_output"]), stride, pad, dil)
    for key, bar in (("output", 5e-5), ("grad_input", 2e-4), ("grad_mask", 2e-4), ("grad_weight", 2e-4), ("grad_bias", 2e-4),
                     ("grad_offset", 2e-4)):
        if got[key] is None or (name == "kink" and key == "grad_offset"):
            continue
        assert got[key].shape == c[key].shape, key
        assert _rel(got[key], c[key]) < bar, (key, _rel(got[key], c[key]))


@needs_fixture
@pytest.mark.parametrize("name", [c[0] for c in DCN1_CASES])
def test_dcn1_hip_extension_equals_reference_extension(name):
    c = _case("dcn1/" + name)
    stride, pad, dil = (int(v) for v in c["geom"])
    got = run_dcn1(hip_conv, torch.from_numpy(c["x"]), torch.from_numpy(c["offset"]), torch.from_numpy(c["weight"]),
                   torch.from_numpy(c["grad_output"]), stride, pad, dil)
    for key, bar in (("output", 5e-5), ("grad_input", 2e-4), ("grad_offset", 2e-4), ("grad_weight_x1p5", 2e-4)):
        assert _rel(got[key], c[key]) < bar, (key, _rel(got[key], c[key]))


@needs_fixture
@pytest.mark.parametrize("i", range(len(POOL_CASES)))
def test_deform_psroi_pooling_hip_extension_equals_reference_extension(i):
    c = _case("pool/%d" % i)
    data, rois, trans, kw = random_case(10 + i, **POOL_CASES[i])
    got = run_pool(hip_pool, data, rois, trans, kw, c["out_grad"])
    assert np.array_equal(got["count"].cpu().numpy(), c["count"]), "sample counts differ"
    assert _rel(got["out"], c["out"]) < 2e-6
    assert _rel(got["data_grad"], c["data_grad"]) < 1e-5
    if not kw["no_trans"]:
        assert _rel(got["trans_grad"], c["trans_grad"]) < 1e-5


# (name, C, H, W, stride): the deformable layers of deformable_resnet50 at 640 x 640 (backbones/resnet.py:113-181,295-309)
REAL_LAYERS = [("layer2.0", 128, 160, 160, 2), ("layer2.1-3", 128, 80, 80, 1), ("layer3.0", 256, 80, 80, 2),
               ("layer3.1-5", 256, 40, 40, 1), ("layer4.0", 512, 40, 40, 2), ("layer4.1-2", 512, 20, 20, 1)]


@pytest.mark.parametrize("layer", REAL_LAYERS, ids=[l[0] for l in REAL_LAYERS])
def test_real_layers_side_by_side_with_the_reference_extension(layer):
    ref_conv, _ = load_reference_extension()
    if ref_conv is None:
        pytest.skip("oracle/_ref did not travel with this snapshot (built where /root/reference exists)")
    name, C, H, W, stride = layer
    g = torch.Generator().manual_seed(C + H)
    N = 2
    oh, ow = H, W                                                  # the offset conv runs at stride 1 (quirk Q10)
    x = torch.randn(N, C, H, W, generator=g)
    om = torch.randn(N, 27, oh, ow, generator=g)
    om[:, :18] = torch.floor(om[:, :18] * 1.5) + 0.25 + 0.5 * torch.rand(N, 18, oh, ow, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * (1.0 / (3.0 * C ** 0.5))
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    gy = torch.randn(N, C, Ho, Wo, generator=g)
    want = run_dcn2(ref_conv, x, om, w, None, gy, stride, 1, 1)
    got = run_dcn2(hip_conv, x, om, w, None, gy, stride, 1, 1)
    for key, bar in (("output", 5e-5), ("grad_input", 2e-4), ("grad_offset", 2e-4), ("grad_mask", 2e-4), ("grad_weight", 2e-4)):
        assert _rel(got[key], want[key]) < bar, (key, _rel(got[key], want[key]))
