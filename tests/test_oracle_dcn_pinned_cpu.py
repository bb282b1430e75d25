"""oracle/dcn.py and oracle/deform_pool.py against golden vectors of the REFERENCE's own extension (assets/ops/dcn/src/*.cu,
*.cpp compiled for gfx950 by oracle/build_ref_ext.sh and run on an MI355X by oracle/gen_golden_dcn.py; fixture
tests/golden/dcn_reference_ext.npz).  This is what pins the two restatements: the reference has no CPU path and no tests of
these ops.  Bars: the reference computes in float32 (im2col + GEMM, float atomics in the backward kernels), the oracles in
float64 -- forward 2e-5 of max|y|, gradients 1e-4 of the tensor's max; PS-RoI pooling sample counts exact."""
import os

import numpy as np
import pytest
import torch

from oracle.dcn import modulated_deform_conv2d
from oracle.deform_pool import psroi_backward, psroi_forward, random_case
from oracle.gen_golden_dcn import DCN1_CASES, DCN2_CASES, POOL_CASES

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "dcn_reference_ext.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(FIXTURE), reason="fixture not generated yet (oracle/gen_golden_dcn.py)")


def _case(prefix):
    z = np.load(FIXTURE)
    return {k[len(prefix) + 1:]: z[k] for k in z.files if k.startswith(prefix + "/")}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("name", [c[0] for c in DCN2_CASES] + ["kink"])
def test_dcn2_oracle_equals_reference_extension(name):
    c = _case("dcn2/" + name)
    stride, pad, dil = (int(v) for v in c["geom"])
    om = torch.from_numpy(c["offset_mask_map"])
    xr = torch.from_numpy(c["x"]).double().requires_grad_(True)
    offr = om[:, :18].double().contiguous().requires_grad_(True)
    mskr = torch.sigmoid(om[:, 18:27]).double().requires_grad_(True)     # the f32 sigmoid the reference was handed
    wr = torch.from_numpy(c["weight"]).double().requires_grad_(True)
    br = torch.from_numpy(c["bias"]).double().requires_grad_(True) if "bias" in c else None
    yr = modulated_deform_conv2d(xr, offr, mskr, wr, br, stride, pad, dil)
    yr.backward(torch.from_numpy(c["grad_output"]).double())
    assert yr.shape == c["output"].shape
    assert _rel(yr.detach(), c["output"]) < 2e-5
    assert _rel(xr.grad, c["grad_input"]) < 1e-4
    assert _rel(mskr.grad, c["grad_mask"]) < 1e-4
    assert _rel(wr.grad, c["grad_weight"]) < 1e-4
    if br is not None:
        assert _rel(br.grad, c["grad_bias"]) < 1e-4
    if name != "kink":     # at integer sample coordinates the offset gradient is one-sided by convention (module docstring)
        assert _rel(offr.grad, c["grad_offset"]) < 1e-4


@pytest.mark.parametrize("name", [c[0] for c in DCN1_CASES])
def test_dcn1_oracle_equals_reference_extension(name):
    c = _case("dcn1/" + name)
    stride, pad, dil = (int(v) for v in c["geom"])
    xr = torch.from_numpy(c["x"]).double().requires_grad_(True)
    offr = torch.from_numpy(c["offset"]).double().requires_grad_(True)
    wr = torch.from_numpy(c["weight"]).double().requires_grad_(True)
    N, _, Ho, Wo = c["offset"].shape
    yr = modulated_deform_conv2d(xr, offr, torch.ones(N, 9, Ho, Wo, dtype=torch.float64), wr, None, stride, pad, dil)
    yr.backward(torch.from_numpy(c["grad_output"]).double())
    assert _rel(yr.detach(), c["output"]) < 2e-5
    assert _rel(xr.grad, c["grad_input"]) < 1e-4
    assert _rel(offr.grad, c["grad_offset"]) < 1e-4
    assert _rel(1.5 * wr.grad, c["grad_weight_x1p5"]) < 1e-4     # accumulated with scale 1 and scale 0.5


@pytest.mark.parametrize("i", range(len(POOL_CASES)))
def test_deform_psroi_pooling_oracle_equals_reference_extension(i):
    c = _case("pool/%d" % i)
    data, rois, trans, kw = random_case(10 + i, **POOL_CASES[i])
    assert np.array_equal(data, c["data"]) and np.array_equal(rois, c["rois"])       # the fixture's inputs are these
    out_o, cnt_o = psroi_forward(data, rois, trans, **kw)
    assert np.array_equal(cnt_o.astype(np.float32), c["count"]), "sample counts differ"
    assert _rel(out_o, c["out"]) < 2e-6
    dg_o, tg_o = psroi_backward(c["out_grad"], data, rois, trans, cnt_o, **kw)
    assert _rel(dg_o, c["data_grad"]) < 1e-5
    if not kw["no_trans"]:
        assert _rel(tg_o, c["trans_grad"]) < 1e-5
