"""Oracle greedy CTC decode -- restates structure/representers/ctc_representer.py:20-34.

Rules: argmax over classes (first index on ties); skip a symbol if it equals the previous
emitted-or-blank symbol OR is `unknown` (an unknown does NOT update `previous`); emit when it is not blank;
then previous = symbol.
"""
import numpy as np


def greedy_decode(pred_nc1t, blank=0, unknown=1):
    """pred_nc1t: array-like [N, C, 1, T] class scores.  Returns int32 [N, T] (blank padded) like the reference."""
    p = np.asarray(pred_nc1t)
    idx = p.argmax(axis=1)[:, 0, :]  # [N, T]
    out = np.full(idx.shape, blank, dtype=np.int32)
    for i in range(idx.shape[0]):
        valid = 0
        previous = blank
        for j in range(idx.shape[1]):
            c = int(idx[i, j])
            if c == previous or c == unknown:
                continue
            if c != blank:
                out[i, valid] = c
                valid += 1
            previous = c
    return out
