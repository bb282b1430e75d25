"""The HIP 2D-CTC kernels (`megreader_amd.ops.ctc_2d.ctc_2d_csrc`: mr_ctc2d_fwd / mr_ctc2d_bwd) against the REFERENCE's own
`ctc_2d_csrc` extension: the fixture leg always runs (tests/golden/ctc2d_reference_ext.npz, recorded on an MI355X from
ops/ctc_2d/csrc/** compiled for gfx950 by oracle/build_ref_ext.sh); the live leg runs both modules side by side at the
published 2D-CTC shape (T = W = 32, H = 8, N = 256, C = 38, S = 32: experiments/.../res50-ppm-2d-ctc.yaml) when oracle/_ref
travelled with the snapshot.  Same calls as ops/ctc_2d/ctc_loss_2d.py:15-35."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from megreader_amd.ops.ctc_2d import ctc_2d_csrc as hip_ext  # noqa: E402
from oracle.gen_golden_ctc2d_ext import CASES, inputs, load_reference_extension, run  # noqa: E402

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "ctc2d_reference_ext.npz")


def _check(got, want_nll, want_alpha, want_grad):
    nll, la, grad = (got[k].detach().float().cpu().numpy() for k in ("nll", "log_alpha", "grad"))
    assert np.abs(nll - want_nll).max() < 2e-5 * max(1.0, np.abs(want_nll).max())
    fin = np.isfinite(want_alpha) & (want_alpha > -1e30)
    assert la.shape == want_alpha.shape
    assert np.abs(la[fin] - want_alpha[fin]).max() < 1e-4 * max(1.0, np.abs(want_alpha[fin]).max())
    scale = max(1e-3, np.abs(want_grad).max())
    assert np.abs(grad - want_grad).max() < 2e-4 * scale
    assert ((want_grad == 0) == (grad == 0)).all()


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="fixture not generated yet (oracle/gen_golden_ctc2d_ext.py)")
@pytest.mark.parametrize("i", range(len(CASES)))
def test_hip_ctc2d_extension_equals_reference_extension_fixture(i):
    z = np.load(FIXTURE)
    c = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("%d/" % i)}
    got = run(hip_ext, c["log_probs"], c["targets"], c["input_lengths"], c["target_lengths"], c["grad_out"])
    _check(got, c["nll"], c["log_alpha"], c["grad"])


def test_published_shape_side_by_side_with_the_reference_extension():
    ref_ext = load_reference_extension()
    if ref_ext is None:
        pytest.skip("oracle/_ref did not travel with this snapshot (built where /root/reference exists)")
    lp, tg, il, tl, go = inputs((32, 8, 256, 38, 32, 12))
    want = run(ref_ext, lp, tg, il, tl, go)
    got = run(hip_ext, lp, tg, il, tl, go)
    _check(got, *(want[k].detach().float().cpu().numpy() for k in ("nll", "log_alpha", "grad")))
