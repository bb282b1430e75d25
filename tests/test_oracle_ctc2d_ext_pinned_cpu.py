"""oracle/ctc2d.py against golden vectors of the REFERENCE's own CUDA 2D-CTC kernels (ops/ctc_2d/csrc/** compiled for gfx950 by
oracle/build_ref_ext.sh, run on an MI355X by oracle/gen_golden_ctc2d_ext.py; tests/golden/ctc2d_reference_ext.npz).  The
restatement was pinned by the reference's PYTHON CTCLoss2D so far (tests/test_oracle_ctc2d.py); this adds the extension the
training path really binds (ops/ctc_2d/ctc_loss_2d.py:4): nll, the saved log_alpha tensor [N, T, H, 2S+1] and the returned
gradient with the collect kernel's conventions (additive exp(lp) term, -inf -> 0 rule, zero rows past input_length)."""
import os

import numpy as np
import pytest

from oracle.ctc2d import ctc2d
from oracle.gen_golden_ctc2d_ext import CASES

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "ctc2d_reference_ext.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(FIXTURE), reason="fixture not generated yet (oracle/gen_golden_ctc2d_ext.py)")


@pytest.mark.parametrize("i", range(len(CASES)))
def test_ctc2d_oracle_equals_reference_cuda_extension(i):
    z = np.load(FIXTURE)
    c = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("%d/" % i)}
    ref = ctc2d(c["log_probs"], c["targets"], c["input_lengths"], c["target_lengths"], blank=0, grad_out=c["grad_out"])
    assert np.abs(ref["nll"] - c["nll"]).max() < 2e-5 * max(1.0, np.abs(ref["nll"]).max())
    # log_alpha: same finite pattern, values to float32 accuracy (entries the reference never writes stay at its fill value;
    # compare where the oracle is finite)
    fin = np.isfinite(ref["alpha"])
    la = c["log_alpha"]
    assert la.shape == ref["alpha"].shape
    assert np.abs(la[fin] - ref["alpha"][fin]).max() < 1e-4 * max(1.0, np.abs(ref["alpha"][fin]).max())
    assert not np.isfinite(la[~fin]).any() or (la[~fin] < -1e30).all()
    scale = max(1e-3, np.abs(ref["grad"]).max())
    assert np.abs(c["grad"] - ref["grad"]).max() < 2e-4 * scale
    assert ((ref["grad"] == 0) == (c["grad"] == 0)).all()       # exact zero pattern
