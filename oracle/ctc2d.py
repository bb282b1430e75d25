"""Oracle restatement (float64, numpy) of the reference's CUDA-only 2D-CTC op `ops/ctc_2d`.

Follows ops/ctc_2d/csrc/cuda/ctc2d_cuda_kernel.cu:
    :33-42   extended target l'_s (blank at even s)
    :55-211  log_alpha recursion and neg_log_likelihood
    :254-368 log_beta recursion
    :427-517 the gradient actually returned (collect kernel; `is_large = 0`), including the rule that a class whose
             collected log(alpha*beta) is -inf gets gradient 0 and every other class gets
             (exp(lp) - exp(G + nll - lp)) * grad_out  -- NOT d nll / d lp  
Caller contract: decoders/ctc_decoder2d.py:37-51 (lp = log(max(mask*classify, tiny)) laid out [W, H, N, C]).

PARITY STATUS: PINNED (forward and gradient) to the reference's own differentiable pure-python implementation
`decoders/ctc_loss2d.py::CTCLoss2D` in its valid (NLL <~ 80) regime -- oracle/gen_golden.py::ctc2d_fixture runs it on a
peaked batch and refuses to write tests/golden/ctc2d_golden.pt unless (a) nll agrees to 2e-5 (f32 class) and (b) the
gradient convention agrees: autograd of the python class w.r.t. log-classify is -occupancy, and
exp(lp) - oracle.grad == occupancy to 1e-6 (float64) wherever the oracle's gradient is non-zero, with the oracle
exactly 0 wherever the reference occupancy is 0 (tests/test_oracle_ctc2d.py::test_gradient_pinned_to_reference_...).
What the python class cannot pin (it has no such term): the additive exp(lp) on extended-target classes and the
"-inf -> 0" rule are the CUDA collect kernel's (ctc2d_cuda_kernel.cu:498-515), restated from the source.
Round 4: ALSO pinned to the CUDA extension itself -- ops/ctc_2d/csrc/** compiles for gfx950 from the sources where they lie
(oracle/build_ref_ext.sh -> oracle/_ref/ctc_2d_csrc) and runs on the MI355X; oracle/gen_golden_ctc2d_ext.py recorded nll, the
saved log_alpha [N, T, H, 2S+1] and the returned gradient of five cases in tests/golden/ctc2d_reference_ext.npz, and
tests/test_oracle_ctc2d_ext_pinned_cpu.py holds this restatement to them (including the exact zero pattern of the gradient,
i.e. the collect kernel's additive term and its "-inf -> 0" rule, which the python class could not pin).
Further anchors: H = 1 == torch.nn.functional.ctc_loss; sum over (h, s) of exp(alpha+beta-lp+nll) == 1 for every t;
finite differences of nll.
"""
import numpy as np

NEG = -np.inf


def _lse(v):
    v = np.asarray(v, dtype=np.float64)
    m = v.max() if v.size else NEG
    if m == NEG:
        return NEG
    return m + np.log(np.exp(v - m).sum())


def ext_targets(targets_b, L, blank):
    ext = [blank]
    for k in range(L):
        ext += [int(targets_b[k]), blank]
    return ext


def ctc2d(lp, targets, input_lengths, target_lengths, blank=0, grad_out=None):
    """lp [T,H,N,C]; targets [N,S]; lengths [N].  Returns dict(nll[N], alpha[N,T,H,2S+1], beta[...], grad[T,H,N,C])."""
    lp = np.asarray(lp, dtype=np.float64)
    T, H, N, C = lp.shape
    S = np.asarray(targets).shape[1]
    SPm = 2 * S + 1
    alpha = np.full((N, T, H, SPm), NEG)
    beta = np.full((N, T, H, SPm), NEG)
    nll = np.zeros(N)
    grad = np.zeros((T, H, N, C))
    go = np.ones(N) if grad_out is None else np.asarray(grad_out, dtype=np.float64)
    for b in range(N):
        Tb, L = int(input_lengths[b]), int(target_lengths[b])
        ext = ext_targets(targets[b], L, blank)
        SP = 2 * L + 1
        # ---- alpha (:84-184)
        alpha[b, 0, :, 0] = lp[0, :, b, blank]
        if L > 0:
            alpha[b, 0, :, 1] = lp[0, :, b, ext[1]]
        for t in range(1, T):
            if not (t < Tb and L > 0):
                continue
            A = [_lse(alpha[b, t - 1, :, s]) for s in range(SP)]
            for s in range(SP):
                terms = [A[s]]
                if s > 0:
                    terms.append(A[s - 1])
                if s > 1 and ext[s] != ext[s - 2]:
                    terms.append(A[s - 2])
                tr = _lse(terms)
                alpha[b, t, :, s] = tr + lp[t, :, b, ext[s]] if tr != NEG else NEG
        # ---- nll (:189-209)
        l1 = _lse(alpha[b, Tb - 1, :, 2 * L])
        l2 = _lse(alpha[b, Tb - 1, :, 2 * L - 1]) if L > 0 else NEG
        nll[b] = -_lse([l1, l2])
        # ---- beta (:282-366)
        beta[b, Tb - 1, :, 2 * L] = lp[Tb - 1, :, b, blank]
        if L > 0:
            beta[b, Tb - 1, :, 2 * L - 1] = lp[Tb - 1, :, b, ext[2 * L - 1]]
        for t in range(Tb - 2, -1, -1):
            if L == 0:
                continue
            Bn = [_lse(beta[b, t + 1, :, s]) for s in range(SP)]
            for s in range(SP):
                terms = [Bn[s]]
                if s < 2 * L:
                    terms.append(Bn[s + 1])
                if s < 2 * L - 1 and ext[s + 2] != ext[s]:
                    terms.append(Bn[s + 2])
                tr = _lse(terms)
                beta[b, t, :, s] = tr + lp[t, :, b, ext[s]] if tr != NEG else NEG
        # ---- gradient (:460-515)
        if L > 0:
            for t in range(Tb):
                for h in range(H):
                    for c in set(ext):
                        G = _lse([alpha[b, t, h, s] + beta[b, t, h, s] for s in range(SP) if ext[s] == c])
                        if G != NEG:
                            x = lp[t, h, b, c]
                            grad[t, h, b, c] = (np.exp(x) - np.exp(G + nll[b] - x)) * go[b]
    return {'nll': nll, 'alpha': alpha, 'beta': beta, 'grad': grad}


def synthetic_lp(T, H, N, C, seed=0, peak=0.0, targets=None, target_lengths=None):
    """log(max(mask*classify, tiny)) like decoders/ctc_decoder2d.py:37-45 from random logits (float32).
    peak > 0 biases the distributions towards a valid alignment of `targets` (keeps NLL small)."""
    rng = np.random.RandomState(seed)
    mask_logit = rng.randn(T, H, N).astype(np.float32)
    cls_logit = rng.randn(T, H, N, C).astype(np.float32)
    if peak > 0 and targets is not None:
        for b in range(N):
            L = int(target_lengths[b])
            for t in range(T):
                k = min(L - 1, t * L // T)
                cls_logit[t, :, b, int(targets[b][k])] += peak
    mask = np.exp(mask_logit - mask_logit.max(axis=1, keepdims=True))
    mask /= mask.sum(axis=1, keepdims=True)
    cls = np.exp(cls_logit - cls_logit.max(axis=3, keepdims=True))
    cls /= cls.sum(axis=3, keepdims=True)
    tiny = np.finfo(np.float32).tiny
    return np.log(np.maximum(mask[..., None] * cls, tiny)).astype(np.float32), np.log(mask), np.log(cls)
