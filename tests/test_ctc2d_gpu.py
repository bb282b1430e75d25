"""GPU parity of the 2D-CTC op (ops.ctc_loss_2d mirror) against the float64 oracle restatement of the reference's
CUDA kernels (oracle/ctc2d.py) and against the golden vectors produced with the reference's own python CTCLoss2D."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from megreader_amd.ops import ctc_loss_2d  # noqa: E402
from oracle.ctc2d import ctc2d, synthetic_lp  # noqa: E402

DEV = "cuda"


def _targets(N, S, C, rng, lmin=1, lmax=6):
    lengths = rng.randint(lmin, lmax + 1, size=N)
    tg = np.zeros((N, S), dtype=np.int64)
    for i, L in enumerate(lengths):
        tg[i, :L] = rng.randint(1, C, size=L)
    return tg, lengths.astype(np.int64)


@pytest.mark.parametrize("T,H,N,C,S", [(12, 4, 3, 10, 8), (16, 4, 5, 38, 32), (9, 1, 4, 7, 5), (20, 8, 2, 38, 32)])
def test_matches_oracle(T, H, N, C, S):
    rng = np.random.RandomState(T * 7 + H)
    tg, tl = _targets(N, S, C, rng, 1, min(6, (T - 1) // 2))
    if N > 2:
        tg[2, 1] = tg[2, 0]  # adjacent repeat (needs a blank in between)
        tl[2] = max(tl[2], 2)
    il = np.full(N, T, dtype=np.int64)
    if N > 1:
        il[1] = T - 3      # ragged input length
    lp, _, _ = synthetic_lp(T, H, N, C, seed=T)
    go = rng.rand(N) + 0.5
    ref = ctc2d(lp, tg, il, tl, blank=0, grad_out=go)
    x = torch.from_numpy(lp).to(DEV).requires_grad_(True)
    nll = ctc_loss_2d(x, torch.from_numpy(tg).to(DEV), torch.from_numpy(il).to(DEV), torch.from_numpy(tl).to(DEV))
    assert nll.shape == (N,) and nll.dtype == torch.float32
    assert np.abs(nll.detach().cpu().numpy() - ref['nll']).max() < 2e-5 * max(1.0, np.abs(ref['nll']).max())
    nll.backward(torch.from_numpy(go).float().to(DEV))
    g = x.grad.cpu().numpy()
    scale = max(1e-3, np.abs(ref['grad']).max())
    assert np.abs(g - ref['grad']).max() < 2e-4 * scale
    # exact zero pattern: classes outside the target and (t >= input_length) rows
    assert ((ref['grad'] == 0) == (g == 0)).all()


def test_h1_equals_torch_ctc():
    T, N, C, S = 15, 4, 12, 6
    rng = np.random.RandomState(3)
    tg, tl = _targets(N, S, C, rng, 1, 5)
    logits = torch.randn(T, N, C, generator=torch.Generator().manual_seed(1))
    lp = torch.log_softmax(logits, dim=2)
    ref = torch.nn.functional.ctc_loss(lp.double(), torch.from_numpy(tg), torch.full((N,), T), torch.from_numpy(tl),
                                       reduction='none')
    nll = ctc_loss_2d(lp.unsqueeze(1).contiguous().to(DEV), torch.from_numpy(tg).to(DEV),
                      torch.full((N,), T, dtype=torch.int64).to(DEV), torch.from_numpy(tl).to(DEV))
    assert float((nll.cpu().double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())


def test_golden_from_reference_python_ctcloss2d(golden_dir):
    g = torch.load(os.path.join(golden_dir, "ctc2d_golden.pt"), weights_only=False)
    x = g['lp'].to(DEV).requires_grad_(True)
    nll = ctc_loss_2d(x, g['targets'].to(DEV), g['input_lengths'].to(DEV), g['target_lengths'].to(DEV))
    assert float((nll.cpu() - g['nll_reference_python']).abs().max()) < 1e-4
    assert float((nll.cpu().double() - g['nll_oracle']).abs().max()) < 1e-4
    nll.backward(torch.ones_like(nll))
    assert float((x.grad.cpu().double() - g['grad_oracle']).abs().max()) < 1e-4
    # the gradient PIN to the reference itself: autograd of the reference's python CTCLoss2D w.r.t. log-classify is
    # -occupancy (oracle/gen_golden.py::ctc2d_fixture, float64); the HIP op returns exp(lp) - occupancy on the
    # extended-target classes with finite log(alpha*beta) and exactly 0 elsewhere (ctc2d_cuda_kernel.cu:498-515)
    gr = x.grad.cpu().double()
    occ_ref = g['occupancy_reference_python']
    nz = gr != 0
    recon = torch.where(nz, torch.exp(g['lp'].double()) - gr, torch.zeros_like(gr))
    err = float((recon - occ_ref).abs().max())
    print("2D-CTC gradient vs the reference's autograd occupancy: max|d| %.2e" % err)
    assert err < 2e-5
    assert float(occ_ref[~nz].abs().max()) == 0.0, "HIP gradient is zero where the reference occupancy is not"
    assert bool((nz == (g['grad_oracle'] != 0)).all()), "zero pattern differs from the pinned oracle"


def test_full_size_properties():
    """BASELINE size (T=32, H=8, N=256, C=38, S=32): occupancy of every valid time step sums to 1."""
    T, H, N, C, S = 32, 8, 256, 38, 32
    rng = np.random.RandomState(0)
    tg, tl = _targets(N, S, C, rng, 3, 10)
    lp, _, _ = synthetic_lp(T, H, N, C, seed=5)
    x = torch.from_numpy(lp).to(DEV).requires_grad_(True)
    il = torch.full((N,), T, dtype=torch.int64, device=DEV)
    nll = ctc_loss_2d(x, torch.from_numpy(tg).to(DEV), il, torch.from_numpy(tl).to(DEV))
    assert torch.isfinite(nll).all() and float(nll.min()) > 0
    nll.backward(torch.ones_like(nll))
    # grad = exp(lp) - occupancy on target classes (G finite) => sum over (h, c in target) of (exp(lp) - grad) == 1
    occ = torch.where(x.grad != 0, torch.exp(x.detach()) - x.grad, torch.zeros_like(x.grad))
    tot = occ.sum(dim=(1, 3))          # [T, N]
    assert float((tot - 1).abs().max()) < 2e-3


def test_errors_like_reference():
    lp = torch.zeros(4, 2, 1, 5)
    with pytest.raises(NotImplementedError):
        ctc_loss_2d(lp, torch.zeros(1, 2, dtype=torch.long), torch.tensor([4]), torch.tensor([1]))
    lpd = torch.zeros(4, 2, 1, 5, device=DEV)
    with pytest.raises(RuntimeError):
        ctc_loss_2d(lpd.permute(1, 0, 2, 3), torch.zeros(1, 2, dtype=torch.long, device=DEV),
                    torch.tensor([2], device=DEV), torch.tensor([1], device=DEV))
    with pytest.raises(RuntimeError):
        ctc_loss_2d(lpd, torch.zeros(1, 2, dtype=torch.long, device=DEV), torch.tensor([4], device=DEV),
                    torch.tensor([1], device=DEV), 7)


def test_extension_level_ctc_2d_csrc_calling_sequence():
    """`ops.ctc_2d.ctc_2d_csrc.ctc2d_forward / ctc2d_backward` (csrc/ctc2d.h:7-43) driven exactly as the reference's
    own Function does (ops/ctc_2d/ctc_loss_2d.py:15-16,30-35; tests/test_b6_dropin_cpu.py shows that file binding this
    module): forward returns (nll, log_alpha), backward consumes them."""
    from megreader_amd.ops.ctc_2d import ctc_2d_csrc
    T, H, N, C, S = 16, 4, 5, 38, 32
    rng = np.random.RandomState(11)
    tg, tl = _targets(N, S, C, rng, 1, 6)
    il = np.full(N, T, dtype=np.int64)
    lp, _, _ = synthetic_lp(T, H, N, C, seed=3)
    go = rng.rand(N) + 0.5
    ref = ctc2d(lp, tg, il, tl, blank=0, grad_out=go)
    x = torch.from_numpy(lp).to(DEV)
    args = (torch.from_numpy(tg).to(DEV), torch.from_numpy(il).to(DEV), torch.from_numpy(tl).to(DEV))
    nll, log_alpha = ctc_2d_csrc.ctc2d_forward(x, *args, 0, torch.finfo().tiny)
    assert tuple(log_alpha.shape) == (N, T, H, 2 * S + 1)
    assert np.abs(nll.cpu().numpy() - ref['nll']).max() < 2e-5 * max(1.0, np.abs(ref['nll']).max())
    grad = ctc_2d_csrc.ctc2d_backward(torch.from_numpy(go).float().to(DEV), x, *args, nll, log_alpha, 0)
    assert grad.shape == x.shape
    assert np.abs(grad.cpu().numpy() - ref['grad']).max() < 2e-4 * max(1e-3, np.abs(ref['grad']).max())
    with pytest.raises(RuntimeError):       # AT_CHECK contiguous (ctc2d_cuda.cu:35)
        ctc_2d_csrc.ctc2d_forward(x.permute(0, 1, 3, 2), *args, 0, 1e-30)
    with pytest.raises(RuntimeError):       # blank out of range (ctc2d_cuda.cu:36)
        ctc_2d_csrc.ctc2d_forward(x, *args, C, 1e-30)
