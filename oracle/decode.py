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


def greedy_decode_2d(classify_nchw, mask_n1hw, blank=0, unknown=1):
    """2D-CTC greedy decode -- restates structure/representers/ctc_representer2d.py:27-51.

    heatmap = classify * mask; per column w pick h* = argmax_h max_c heatmap[n, c, h, w] (first index on ties), then
    c* = argmax_c heatmap[n, c, h*, w] (first index on ties); collapse like the 1-D rule.  Returns int32 [N, W]."""
    cl = np.asarray(classify_nchw)
    mk = np.asarray(mask_n1hw)
    heat = cl * mk                                     # (N, C, H, W), same dtype arithmetic as the reference
    hstar = heat.max(axis=1).argmax(axis=1)            # (N, W)
    n_idx = np.arange(heat.shape[0])[:, None]
    w_idx = np.arange(heat.shape[3])[None, :]
    sel = heat[n_idx, :, hstar, w_idx]                 # (N, W, C)
    idx = sel.argmax(axis=2)                           # (N, W)
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


ENGLISH = [None, None] + list("0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ")   # concern/charsets.py:25-27,104-107 (Q6)


def label_to_string(label, charset=ENGLISH, blank=0, unknown=1):
    """concern/charsets.py:60-62: drop blank/unknown ids, map the rest through the charset."""
    return "".join(charset[int(i)] for i in label if int(i) not in (unknown, blank))


def levenshtein(a, b):
    """editdistance.eval (third-party, unpinned in requirement.txt): the standard unit-cost edit distance."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def measure(label_ids, pred_ids, charset=ENGLISH):
    """structure/measurers/sequence_recognition_measurer.py:66-72,101-112 on id arrays [N, S]:
    accuracy[i] = (label_string.upper() == pred_string.upper());
    edit_distance[i] = 0 if len(label) == 0 else 1 - min(len(label), ed(label, pred)) / len(label)."""
    acc, eds = [], []
    for lab, pred in zip(np.asarray(label_ids), np.asarray(pred_ids)):
        ls = label_to_string(lab, charset).upper()
        ps = label_to_string(pred, charset).upper()
        acc.append(ls == ps)
        eds.append(0.0 if len(ls) == 0 else float(1 - min(len(ls), levenshtein(ls, ps)) * 1.0 / len(ls)))
    return acc, eds
