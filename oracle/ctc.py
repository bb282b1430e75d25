"""Explicit float64 restatement of the 1-D CTC alpha/beta recursions and of the gradient w.r.t. the LOGITS
(the reference calls torch's nn.CTCLoss at decoders/crnn.py:48,96-98).

Used to check the HIP kernel's intermediate quantities (log-probs, alpha, per-sample nll, logits gradient) and
pinned against torch.nn.functional.ctc_loss in tests/test_oracle.py.  numpy only.
"""
import numpy as np

NEG = -np.inf


def _lse(*xs):
    m = max(xs)
    if m == NEG:
        return NEG
    return m + np.log(sum(np.exp(x - m) for x in xs))


def ctc_1d(logits, targets, target_lengths, blank=0, zero_infinity=True):
    """logits [T,N,C] (any float dtype), targets [N,S] ints, target_lengths [N].
    Returns dict(loss, nll[N], log_probs[T,N,C] f32-rounded, grad_logits[T,N,C]) for reduction='mean', grad_out=1."""
    x = np.asarray(logits, dtype=np.float32)
    T, N, C = x.shape
    mx = x.max(axis=2, keepdims=True)
    lp32 = (x - (mx + np.log(np.exp(x - mx).sum(axis=2, keepdims=True)))).astype(np.float32)
    lp = lp32.astype(np.float64)
    nll = np.zeros(N)
    grad = np.zeros((T, N, C))
    for b in range(N):
        L = int(target_lengths[b])
        ext = [blank]
        for k in range(L):
            ext += [int(targets[b][k]), blank]
        SP = len(ext)
        al = np.full((T, SP), NEG)
        al[0, 0] = lp[0, b, blank]
        if SP > 1:
            al[0, 1] = lp[0, b, ext[1]]
        for t in range(1, T):
            for s in range(SP):
                a = [al[t - 1, s]]
                if s > 0:
                    a.append(al[t - 1, s - 1])
                if s > 1 and ext[s] != ext[s - 2]:
                    a.append(al[t - 1, s - 2])
                v = _lse(*a)
                al[t, s] = v + lp[t, b, ext[s]] if v != NEG else NEG
        ll = _lse(al[T - 1, SP - 1], al[T - 1, SP - 2]) if SP > 1 else al[T - 1, 0]
        nll[b] = -ll
        if nll[b] == np.inf and zero_infinity:
            continue
        be = np.full((T, SP), NEG)
        be[T - 1, SP - 1] = lp[T - 1, b, blank]
        if SP > 1:
            be[T - 1, SP - 2] = lp[T - 1, b, ext[SP - 2]]
        for t in range(T - 2, -1, -1):
            for s in range(SP):
                a = [be[t + 1, s]]
                if s + 1 < SP:
                    a.append(be[t + 1, s + 1])
                if s + 2 < SP and ext[s] != ext[s + 2]:
                    a.append(be[t + 1, s + 2])
                v = _lse(*a)
                be[t, s] = v + lp[t, b, ext[s]] if v != NEG else NEG
        k = 1.0 / (N * max(L, 1))
        for t in range(T):
            occ = np.zeros(C)
            for c in set(ext):
                terms = [al[t, s] + be[t, s] for s in range(SP) if ext[s] == c]
                g = _lse(*terms)
                if g != NEG:
                    occ[c] = np.exp(g + nll[b] - lp[t, b, c])
            grad[t, b] = (np.exp(lp[t, b]) - occ) * k
    per = np.where(np.isinf(nll), 0.0, nll) if zero_infinity else nll
    lens = np.maximum(np.asarray(target_lengths, dtype=np.float64), 1.0)
    return {'loss': float((per / lens).mean()), 'nll': nll, 'log_probs': lp32, 'grad_logits': grad}
