"""Evaluation decode and metrics on the GPU: the reference does these in Python loops over every
sample and time step on the host (structure/representers/ctc_representer.py:20-34, ctc_representer2d.py:27-51,
structure/measurers/sequence_recognition_measurer.py:66-112).  Same rules, bit-exact results
(tests/test_decode_gpu.py against oracle/decode.py, which is pinned to the reference)."""
import torch

from .._lib import call, ptr, require_cuda

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float64: 2}


def ctc_greedy_decode(pred, blank=0, unknown=1):
    """pred: [N, C, 1, T] or [N, C, T] class scores on the GPU (f32 / bf16 / f64, any strides).
    Returns (ids i32 [N, T] blank padded, lengths i32 [N])."""
    require_cuda(pred)
    if pred.dim() == 4:
        if pred.shape[2] != 1:
            raise RuntimeError("ctc_greedy_decode expects [N, C, 1, T]")
        pred = pred.select(2, 0)           # CTCRepresenter: pred.select(1, 0) after the arg-max over C
    if pred.dtype not in _DT:
        raise TypeError("ctc_greedy_decode: unsupported dtype %s" % pred.dtype)
    N, C, T = pred.shape
    out = torch.empty((N, T), dtype=torch.int32, device=pred.device)
    lengths = torch.empty((N,), dtype=torch.int32, device=pred.device)
    sn, sc, st = pred.stride()
    call("mr_ctc_greedy_decode", _DT[pred.dtype], ptr(pred), sn, sc, st, N, C, T, int(blank), int(unknown), ptr(out),
         ptr(lengths))
    return out, lengths


def ctc2d_greedy_decode(classify, mask, blank=0, unknown=1):
    """classify [N, C, H, W], mask [N, 1, H, W] on the GPU (any strides; converted to f32 if needed).
    Returns (ids i32 [N, W], lengths i32 [N])."""
    require_cuda(classify, mask)
    if classify.dtype != torch.float32:
        classify = classify.float()
    if mask.dtype != torch.float32:
        mask = mask.float()
    N, C, H, W = classify.shape
    if tuple(mask.shape) != (N, 1, H, W):
        raise RuntimeError("mask must be [N, 1, H, W]")
    out = torch.empty((N, W), dtype=torch.int32, device=classify.device)
    lengths = torch.empty((N,), dtype=torch.int32, device=classify.device)
    cn, cc, ch, cw = classify.stride()
    mn, _, mh, mw = mask.stride()
    call("mr_ctc2d_greedy_decode", ptr(classify), cn, cc, ch, cw, ptr(mask), mn, mh, mw, N, C, H, W, int(blank),
         int(unknown), ptr(out), ptr(lengths))
    return out, lengths


def sequence_measure(labels, preds, blank=0, unknown=1, fold=None):
    """labels i32 [N, S], preds i32 [N, S2] (device).  Returns dict of device tensors:
    accuracy (bool [N]), edit_distance (f64 [N], the reference's normalised score), distance (i32 [N]), label_length."""
    require_cuda(labels, preds)
    labels = labels.to(torch.int32).contiguous()
    preds = preds.to(torch.int32).contiguous()
    N, S = labels.shape
    dev = labels.device
    acc = torch.empty((N,), dtype=torch.int32, device=dev)
    ed = torch.empty((N,), dtype=torch.int32, device=dev)
    ll = torch.empty((N,), dtype=torch.int32, device=dev)
    score = torch.empty((N,), dtype=torch.float64, device=dev)
    if fold is not None:
        fold = fold.to(device=dev, dtype=torch.int32).contiguous()
    call("mr_seq_measure", ptr(labels), S, ptr(preds), preds.shape[1], N, int(blank), int(unknown), ptr(fold), ptr(acc),
         ptr(ed), ptr(ll), ptr(score))
    return {'accuracy': acc.bool(), 'edit_distance': score, 'distance': ed, 'label_length': ll}
