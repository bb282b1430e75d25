"""CPU pins of the 2D-CTC oracle (the reference has no CPU implementation of this CUDA op): H = 1 equals torch's
CTC; the committed golden (produced with the reference's pure-python CTCLoss2D) is reproduced; occupancies sum to 1;
the returned gradient equals autograd's  d nll / d lp  plus the exp(lp) term on target classes."""
import os

import numpy as np
import torch

from oracle.ctc2d import ctc2d, synthetic_lp


def _case(T=10, H=3, N=3, C=9, S=5, seed=0):
    rng = np.random.RandomState(seed)
    tl = rng.randint(1, 4, size=N).astype(np.int64)
    tg = np.zeros((N, S), dtype=np.int64)
    for i, L in enumerate(tl):
        tg[i, :L] = rng.randint(1, C, size=L)
    il = np.full(N, T, dtype=np.int64)
    lp, _, _ = synthetic_lp(T, H, N, C, seed=seed + 1)
    return lp, tg, il, tl


def test_h1_equals_torch_ctc():
    T, N, C, S = 12, 4, 8, 5
    rng = np.random.RandomState(1)
    tl = rng.randint(1, 5, size=N).astype(np.int64)
    tg = np.zeros((N, S), dtype=np.int64)
    for i, L in enumerate(tl):
        tg[i, :L] = rng.randint(1, C, size=L)
    lp = torch.log_softmax(torch.randn(T, N, C, generator=torch.Generator().manual_seed(2)).double(), dim=2)
    ref = torch.nn.functional.ctc_loss(lp, torch.from_numpy(tg), torch.full((N,), T), torch.from_numpy(tl),
                                       reduction='none')
    o = ctc2d(lp.unsqueeze(1).numpy(), tg, np.full(N, T), tl)
    assert np.abs(o['nll'] - ref.numpy()).max() < 1e-10


def test_golden_from_reference_python(golden_dir):
    g = torch.load(os.path.join(golden_dir, "ctc2d_golden.pt"), weights_only=False)
    o = ctc2d(g['lp'].numpy(), g['targets'].numpy(), g['input_lengths'].numpy(), g['target_lengths'].numpy())
    assert np.abs(o['nll'] - g['nll_reference_python'].numpy()).max() < 2e-5
    assert np.abs(o['nll'] - g['nll_oracle'].numpy()).max() < 1e-12
    assert np.abs(o['grad'] - g['grad_oracle'].numpy()).max() < 1e-12


def test_gradient_pinned_to_reference_python_autograd(golden_dir):
    """The gradient PIN: autograd of the reference's differentiable python CTCLoss2D (decoders/ctc_loss2d.py:86-154,
    run in float64 by oracle/gen_golden.py) w.r.t. log-classify is -occupancy; the CUDA op returns
    exp(lp) - occupancy on the extended-target classes whose log(alpha*beta) is finite and 0 elsewhere
    (ctc2d_cuda_kernel.cu:498-515).  The oracle's gradient must reproduce the reference occupancy element-wise."""
    g = torch.load(os.path.join(golden_dir, "ctc2d_golden.pt"), weights_only=False)
    lp = g['lp'].numpy()
    tg, tl = g['targets'].numpy(), g['target_lengths'].numpy()
    o = ctc2d(lp, tg, g['input_lengths'].numpy(), tl)
    occ = g['occupancy_reference_python'].numpy()
    nz = o['grad'] != 0
    recon = np.where(nz, np.exp(lp.astype(np.float64)) - o['grad'], 0.0)
    assert np.abs(recon - occ).max() < 1e-6
    assert np.abs(occ[~nz]).max() == 0.0                 # the oracle is zero exactly where the reference occupancy is
    assert np.abs(recon.sum(axis=3) - g['mask_occupancy_reference_python'].numpy()).max() < 1e-6
    # classes outside the extended target: occupancy 0 in the reference, gradient 0 in the op ("0 elsewhere")
    N, C = lp.shape[2], lp.shape[3]
    for b in range(N):
        ext = set([0] + [int(v) for v in tg[b, :int(tl[b])]])
        for c in range(C):
            if c not in ext:
                assert not nz[:, :, b, c].any() and np.abs(occ[:, :, b, c]).max() == 0.0
    assert occ.max() > 0.5 and nz.mean() > 0.1           # the batch is peaked: the pin is not vacuous


def test_occupancy_sums_to_one_and_gradient_convention():
    lp, tg, il, tl = _case()
    o = ctc2d(lp, tg, il, tl)
    T, H, N, C = lp.shape
    lp64 = lp.astype(np.float64)
    for b in range(N):
        tot = np.zeros(T)
        for t in range(T):
            ab = o['alpha'][b, t] + o['beta'][b, t]          # [H, S']
            SP = 2 * int(tl[b]) + 1
            ext = [0] + [v for k in range(int(tl[b])) for v in (int(tg[b, k]), 0)]
            for h in range(H):
                for s in range(SP):
                    if np.isfinite(ab[h, s]):
                        tot[t] += np.exp(ab[h, s] - lp64[t, h, b, ext[s]] + o['nll'][b])
        assert np.abs(tot - 1).max() < 1e-9
    # finite-difference check of d nll / d lp on a few entries: grad_returned - exp(lp) == d nll / d lp where G finite
    eps = 1e-6
    rng = np.random.RandomState(5)
    for _ in range(6):
        t, h, b = rng.randint(T), rng.randint(H), rng.randint(N)
        c = int(tg[b, 0])
        lp2 = lp64.copy()
        lp2[t, h, b, c] += eps
        d = (ctc2d(lp2, tg, il, tl)['nll'][b] - o['nll'][b]) / eps
        if o['grad'][t, h, b, c] != 0:
            assert abs((o['grad'][t, h, b, c] - np.exp(lp64[t, h, b, c])) - d) < 1e-4
