"""Attention (Bahdanau) GRU recognition head on HIP kernels.

Mirror of reference decoders/attention_decoder.py:10-231: same constructor, same module / parameter names
(`encode.*`, `decoder.{embedding,word_linear,attn.attn,attn.v,rnn,out}`, `onehot_embedding_{x,y}`), same default
initialisation order, same forward contract (training: `(loss[N], attention[N, max_size, height, max_size])`,
eval: `pred[N, max_size] int32`), same teacher-forcing coin (`np.random.rand() < 0.5` unless `gt_as_output` is set,
quirk Q16) and the same loss mask `timestep <= lengths`.

What runs where: the 7-conv encoder uses the HIP conv/BN/pool layers.  TRAINING runs the 32 decode steps inside ONE autograd
Function (`_DecodeLoopFn`, round 3): per step forward
  [GEMM h -> (hproj | gh)] -> attn_fwd2 -> [GEMM context -> gi_c] -> gru_fwd2 (word part gathered from a table) -> [GEMM out] -> nll
(6 launches; 11 before) and 6 launches backward (about 20 plus autograd's ATen adds before): the word path
embedding -> word_linear -> W_ih[:, :H] only depends on the class index, so it is a [classes, 3H] table computed once per
forward; W_attn[:, :H] and W_hh share their input and are one stacked GEMM; every weight gradient is ONE transpose-read
GEMM over all 32 steps after the loop (per-step inputs / output gradients live in [S, N, .] buffers that the step kernels
write in place); the encoder-side attention gradient is one kernel after the loop.  Eval (greedy decode with early stop)
keeps the per-step path:
  [GEMM word] -> [GEMM hproj] -> attn_step -> [GEMMs gi_w, gi_c, gh] -> gru_gates -> [GEMM out] -> nll_step
with the GEMMs on the MFMA NT/TN kernels.  The reference's per-step Linear(1057 -> 512) over cat([hidden x T, enc])
is split algebraically into hidden and encoder halves, so the encoder half (eproj) is one GEMM per sequence.
Weights shared by the 32 steps are converted once per forward; their gradients (and those of eproj / enc / v)
accumulate in f32 buffers across the steps and are handed to autograd by the step that runs backward last (step 0).
Glue that is pure data movement on tiny tensors (one-hot position embeddings, cat/pad of the encoder sequence) uses
torch ops.
"""
import os

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import get_compute_dtype
from .._lib import call, dtype_code, load, ptr, vec_of
from ..charsets import DefaultCharset
from ..nn import Conv2d, BatchNorm2d, MaxPool2d, FusedReLU
from ..nn import prep


# round 4: GRU kernels in the epilogue of the neighbouring skinny GEMM, output layer + NLL as one kernel (csrc/gemm_skinny.hip);
# MEGREADER_DECODE_FUSED=0 keeps the round-3 chain (A/B)
FUSED_STEP = os.environ.get("MEGREADER_DECODE_FUSED", "1") != "0"
# round 5: with teacher forcing FIXED ON (`gt_as_output=True`, what SURVEY's C4 and bench.py run) no step feeds its arg-max back,
# so the output layer + log-softmax + NLL is not on the recurrence: ONE [S*N, H] x [H, C] GEMM and ONE log-softmax / NLL launch
# behind the loop instead of a launch per step on the chain (32 launches, ~0.23 ms of the FPN step).  MEGREADER_DECODE_BATCHED_OUT=0
# keeps the per-step launches (A/B); random coins (the YAML default) and `gt_as_output=False` always take the per-step path.
BATCHED_OUT = os.environ.get("MEGREADER_DECODE_BATCHED_OUT", "1") != "0"
# round 6: the teacher-forced forward loop and the backward loop as one persistent launch each (mr_decode_persist_fwd / _bwd; also
# mr_tuning.decode_persist).  MEGREADER_DECODE_PERSIST=0 keeps the three launches per step; "fwd" / "bwd" switch one side on.
_PERSIST_ENV = os.environ.get("MEGREADER_DECODE_PERSIST", "1")       # "0" | "1" | "fwd" | "bwd"
PERSIST = _PERSIST_ENV in ("1", "fwd")
PERSIST_BWD = _PERSIST_ENV in ("1", "bwd")


def _persist_workspace(N, dev, backward=False):
    """(exchange buffer, size argument) of mr_decode_persist_fwd / _bwd: zero at launch -- from the pre-zeroed arena (size passed NEGATIVE)
    or, when that is exhausted, from torch's allocator (the C call zeroes it).  The status word (last 256 bytes) joins the list
    nn.functional.LSTM_STATUS collects for tests and bench.py."""
    from ..nn import functional as F_
    nbytes = load().mr_decode_persist_bwd_ws_bytes(N) if backward else load().mr_decode_persist_ws_bytes(N)
    size = nbytes
    arena = F_.ZeroArena.take(dev, (nbytes + 7) // 8)
    if arena is not None:
        ws = arena.view(torch.uint8)[:nbytes]
        size = -nbytes
    else:
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    if F_.LSTM_STATUS is not None:
        F_.LSTM_STATUS.append(ws[nbytes - 256:nbytes - 252])
    return ws, size


def _ceil_to(x, m):
    return (x + m - 1) // m * m


def _owner(t):
    """The Parameter a (possibly sliced) weight lives in, or None."""
    base = t._base if t._base is not None else t
    return base if isinstance(base, nn.Parameter) else None


class _SeqLinear(object):
    """y = x W^T + b for a weight shared by all steps of one forward pass.  `weight` may be a column slice of a
    parameter; K is zero-padded to `kp` (x must already be kp wide).  `weight` / `bias` may also be LISTS of equally wide
    blocks that are stacked along the output dimension (a None bias block is zeros): cat([W_attn[:, :H], W_hh]) of the decode
    loop.  The compute-dtype images (w_n [nout, kp], w_t [kp, np_]) live in the prepared-image cache of the owning
    parameters (nn/prep.py): built on the first use, regenerated by the fused optimizers' ONE batch launch after each
    update -- not by zero fills + conversion launches in every forward pass (12 + 6 launches per training step before)."""

    def __init__(self, weight, bias, kp, dtype):
        blocks = list(weight) if isinstance(weight, (list, tuple)) else [weight]
        bias_blocks = list(bias) if isinstance(bias, (list, tuple)) else [bias]
        stacked = len(blocks) > 1
        self.weight, self.bias = (None, None) if stacked else (weight, bias)
        self.dtype = dtype
        v = vec_of(dtype)
        self.k = blocks[0].shape[1]
        self.nout = sum(w.shape[0] for w in blocks)
        self.kp = kp
        self.np_ = _ceil_to(self.nout, v)
        dev = blocks[0].device
        es = 2 if dtype == torch.bfloat16 else 4
        dets = [w.detach() for w in blocks]
        assert all(w.stride(1) == 1 and w.shape[1] == self.k for w in dets) and kp % v == 0 and kp >= self.k
        assert len(bias_blocks) == len(blocks)
        nout, np_, k = self.nout, self.np_, self.k
        need_bias_image = stacked and any(b is not None for b in bias_blocks)

        def build(old):
            if old is None:
                w_n = torch.zeros((nout, kp), dtype=dtype, device=dev)
                w_t = torch.zeros((kp, np_), dtype=dtype, device=dev)
                b_d = torch.zeros((nout,), dtype=torch.float32, device=dev) if need_bias_image else None
            else:
                w_n, w_t, b_d = old
            jobs, row = [], 0
            for w, b in zip(dets, bias_blocks):
                jobs.append(prep.matrix_job(ptr(w), w.stride(0), ptr(w_n) + row * kp * es, kp, ptr(w_t) + row * es, np_,
                                            w.shape[0], k, 0))
                if need_bias_image and b is not None:
                    jobs.append(prep.bias_job(ptr(b.detach()), 0, ptr(b_d) + row * 4, w.shape[0], 0))
                row += w.shape[0]
            return (w_n, w_t, b_d), jobs

        owners = [_owner(w) for w in blocks] + [_owner(b) for b in bias_blocks if (need_bias_image and b is not None)]
        if all(o is not None for o in owners) and all(b is None or b.dtype == torch.float32 for b in bias_blocks):
            key = ("seqlinear", kp, tuple((w.storage_offset(), tuple(w.shape), w.stride(0)) for w in dets), need_bias_image)
            self.w_n, self.w_t, b_d = prep.prepared(tuple(owners), key, build, dtype)
        else:       # weights that are not (views of) parameters: converted on every call, as before round 5
            (self.w_n, self.w_t, b_d), jobs = build(None)
            for j in jobs:
                prep.run_job(dtype_code(dtype), j)
        if need_bias_image:
            self.bias_d = b_d
        else:
            self.bias_d = bias_blocks[0].detach() if (not stacked and bias_blocks[0] is not None) else None
        self.gw = None
        self.gb = None
        self.calls = 0

    def __call__(self, x):
        first = self.calls == 0
        self.calls += 1
        return _SeqLinearFn.apply(x, self.weight, self.bias, self, first)


class _SeqLinearFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, seq, first):
        dt = dtype_code(seq.dtype)
        M = x.shape[0]
        assert x.dim() == 2 and x.shape[1] == seq.kp and x.is_contiguous() and x.dtype == seq.dtype
        y = torch.empty((M, seq.np_), dtype=seq.dtype, device=x.device)
        if seq.np_ != seq.nout:
            y[:, seq.nout:].zero_()
        call("mr_gemm_nt", dt, ptr(x), seq.kp, ptr(seq.w_n), seq.kp, ptr(y), seq.np_, ptr(seq.bias_d), 0, M, seq.nout,
             seq.kp)
        ctx.save_for_backward(x)
        ctx.seq, ctx.first = seq, first
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        seq = ctx.seq
        dt = dtype_code(seq.dtype)
        M = x.shape[0]
        g = gy if (gy.is_contiguous() and gy.dtype == seq.dtype) else gy.to(seq.dtype).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, seq.kp), dtype=seq.dtype, device=x.device)
            call("mr_gemm_nt", dt, ptr(g), seq.np_, ptr(seq.w_t), seq.np_, ptr(dx), seq.kp, 0, 0, M, seq.kp, seq.np_)
        if seq.gw is None:
            seq.gw = torch.zeros((seq.np_, seq.kp), dtype=torch.float32, device=x.device)
            seq.gb = torch.zeros((seq.np_,), dtype=torch.float32, device=x.device)
        call("mr_gemm_tn", dt, ptr(g), seq.np_, ptr(x), seq.kp, ptr(seq.gw), seq.kp, M, seq.np_, seq.kp, 0,
             ptr(seq.gb) if seq.bias is not None else 0)
        gw = gb = None
        if ctx.first:  # the first forward call runs its backward last: every step has accumulated by now
            gw = seq.gw[:seq.nout, :seq.k]
            gb = seq.gb[:seq.nout] if seq.bias is not None else None
        return dx, gw, gb, None, None


class _AttnStepFn(Function):
    @staticmethod
    def forward(ctx, hproj, eproj, enc, v, state, first):
        dtype = state['dtype']
        N, Tn, Hd = eproj.shape
        Ep = enc.shape[2]
        w = torch.empty((N, Tn), dtype=torch.float32, device=hproj.device)
        context = torch.empty((N, Ep), dtype=dtype, device=hproj.device)
        call("mr_attn_step_fwd", dtype_code(dtype), ptr(hproj), ptr(eproj), ptr(v), ptr(enc), ptr(w), ptr(context), N,
             Tn, Hd, Ep)
        ctx.save_for_backward(hproj, eproj, enc, v, w)
        ctx.meta = (state, first, N, Tn, Hd, Ep, dtype)
        return w, context

    @staticmethod
    def backward(ctx, gw, gcontext):
        hproj, eproj, enc, v, w = ctx.saved_tensors
        state, first, N, Tn, Hd, Ep, dtype = ctx.meta
        dev = hproj.device
        if 'deproj' not in state:
            state['deproj'] = torch.zeros((N, Tn, Hd), dtype=torch.float32, device=dev)
            state['denc'] = torch.zeros((N, Tn, Ep), dtype=torch.float32, device=dev)
            state['dv'] = torch.zeros((Hd,), dtype=torch.float32, device=dev)
        gcontext = gcontext.to(dtype).contiguous()
        gwp = gw.to(torch.float32).contiguous() if gw is not None else None
        dh = torch.empty((N, Hd), dtype=dtype, device=dev)
        call("mr_attn_step_bwd", dtype_code(dtype), ptr(gcontext), ptr(gwp), ptr(hproj), ptr(eproj), ptr(v), ptr(enc),
             ptr(w), ptr(dh), ptr(state['deproj']), ptr(state['dv']), ptr(state['denc']), N, Tn, Hd, Ep)
        if first:
            return dh, state['deproj'].to(dtype), state['denc'].to(dtype), state['dv'], None, None
        return dh, None, None, None, None, None


class _GruGatesFn(Function):
    @staticmethod
    def forward(ctx, gi_a, gi_b, gh, h, dtype):
        N, H = h.shape
        hnew = torch.empty_like(h)
        save = torch.empty((N, 3 * H), dtype=torch.float32, device=h.device)
        call("mr_gru_gates_fwd", dtype_code(dtype), ptr(gi_a), ptr(gi_b), ptr(gh), ptr(h), ptr(hnew), ptr(save), N, H)
        ctx.save_for_backward(save, gh, h)
        ctx.dtype = dtype
        return hnew

    @staticmethod
    def backward(ctx, g):
        save, gh, h = ctx.saved_tensors
        N, H = h.shape
        dtype = ctx.dtype
        g = g.to(dtype).contiguous()
        dgi = torch.empty((N, 3 * H), dtype=dtype, device=h.device)
        dgh = torch.empty((N, 3 * H), dtype=dtype, device=h.device)
        dh = torch.empty_like(h)
        call("mr_gru_gates_bwd", dtype_code(dtype), ptr(g), ptr(save), ptr(gh), ptr(h), ptr(dgi), ptr(dgh), ptr(dh), N,
             H)
        return dgi, dgi, dgh, dh, None


class _EmbedRowsFn(Function):
    """emb = embedding(word_idx) as the padded compute-dtype input of word_linear (attention_decoder.py:187-193, 203):
    one gather kernel forward, one scatter-add (f32 atomics into the table gradient) backward."""

    @staticmethod
    def forward(ctx, idx, table, ldo, dtype):
        idx = idx.to(torch.int64).contiguous()
        tab = table if (table.dtype == torch.float32 and table.is_contiguous()) else table.float().contiguous()
        N, (V, D) = idx.shape[0], tab.shape
        out = torch.empty((N, ldo), dtype=dtype, device=idx.device)
        call("mr_embed_rows_fwd", dtype_code(dtype), ptr(idx), ptr(tab), ptr(out), N, V, D, ldo)
        ctx.save_for_backward(idx)
        ctx.meta = (V, D, ldo, dtype, table.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        V, D, ldo, dtype, tdtype = ctx.meta
        if g.dtype != dtype or not g.is_contiguous():
            g = g.to(dtype).contiguous()
        dtab = torch.zeros((V, D), dtype=torch.float32, device=g.device)
        call("mr_embed_rows_bwd", dtype_code(dtype), ptr(idx), ptr(g), ptr(dtab), idx.shape[0], V, D, ldo)
        return None, dtab.to(tdtype), None, None


class _NllStepFn(Function):
    """loss[n] = NLLLoss(log_softmax(logits), target)[n] * mask[n]; also returns argmax (attention_decoder.py:95-106)."""

    @staticmethod
    def forward(ctx, logits, target, mask, C):
        N = logits.shape[0]
        dev = logits.device
        lp = torch.empty((N, C), dtype=torch.float32, device=dev)
        loss = torch.empty((N,), dtype=torch.float32, device=dev)
        am = torch.empty((N,), dtype=torch.int64, device=dev)
        call("mr_nll_step_fwd", dtype_code(logits.dtype), ptr(logits), logits.stride(0), ptr(target), target.stride(0),
             ptr(mask), ptr(lp), ptr(loss), ptr(am), N, C, 0, 0)
        ctx.save_for_backward(lp, target, mask)
        ctx.meta = (N, C, logits.shape[1], logits.dtype)
        ctx.mark_non_differentiable(am)
        return loss, am

    @staticmethod
    def backward(ctx, gloss, _gam):
        lp, target, mask = ctx.saved_tensors
        N, C, ld, dtype = ctx.meta
        gl = gloss.to(torch.float32).contiguous()
        d = torch.zeros((N, ld), dtype=dtype, device=lp.device)
        call("mr_nll_step_bwd", dtype_code(dtype), ptr(gl), ptr(lp), ptr(target), target.stride(0), ptr(mask), ptr(d),
             ld, N, C)
        return d, None, None, None


class _DecodeLoopFn(Function):
    """All `S` steps of AttentionRNNCell (reference decoders/attention_decoder.py:84-118, 146-231) forward and backward.

    inputs : G [C, 3H] word table (dtype), eproj [N,T,H], enc [N,T,Ep] (dtype), v [H], w_ah = attn.attn.weight[:, :H],
             w_hh, b_hh, w_ic = rnn.weight_ih[:, H:H+E], w_out, b_out, targets_t [S,N] i64, lengths [N], flags (DEVICE int32
             [S]: teacher forcing per step -- read by the step kernels, so a replayed hipGraph follows the coins of the
             current step), meta = (dtype, C, blank)
    outputs: loss [N] f32 = sum_s NLL_s * (s <= length);  attention [N, S, T] f32"""

    @staticmethod
    def forward(ctx, G, eproj, enc, v, w_ah, w_hh, b_hh, w_ic, w_out, b_out, targets_t, lengths, flags, meta):
        dtype, C, blank = meta[:3]
        all_teacher = len(meta) > 3 and bool(meta[3]) and BATCHED_OUT       # host-known: every step is fed its target
        dt = dtype_code(dtype)
        es = 2 if dtype == torch.bfloat16 else 4
        N, T, Hd = eproj.shape
        Ep = enc.shape[2]
        S = targets_t.shape[0]
        dev = enc.device
        H3 = 3 * Hd
        HC = Hd + H3
        cat = _SeqLinear([w_ah, w_hh], [None, b_hh], Hd, dtype)
        ic = _SeqLinear(w_ic.detach(), None, Ep, dtype)
        out = _SeqLinear(w_out.detach(), b_out.detach(), Hd, dtype)
        assert cat.np_ == HC and ic.np_ == H3 and G.shape[1] >= H3 and G.is_contiguous() and G.dtype == dtype
        eproj = eproj if eproj.is_contiguous() else eproj.contiguous()
        vf = v.detach().float().contiguous()
        H_all = torch.empty((S + 1, N, Hd), dtype=dtype, device=dev)
        H_all[0].zero_()
        HC_all = torch.empty((S, N, HC), dtype=dtype, device=dev)
        W_att = torch.empty((S, N, T), dtype=torch.float32, device=dev)
        CTX_all = torch.empty((S, N, Ep), dtype=dtype, device=dev)
        SAVE_all = torch.empty((S, N, H3), dtype=torch.float32, device=dev)
        LP_all = torch.empty((S, N, C), dtype=torch.float32, device=dev)
        gic = torch.empty((N, H3), dtype=dtype, device=dev)
        logits = torch.zeros((N, out.np_), dtype=dtype, device=dev)
        loss = torch.empty((N,), dtype=torch.float32, device=dev)
        am_all = torch.empty((S, N), dtype=torch.int64, device=dev)
        idx_all = None
        mask_all = (torch.arange(S, device=dev).view(S, 1) <= lengths.view(1, N)).to(torch.float32).contiguous()
        ldG = G.shape[1]
        # the word fed to step s + 1 is the target of step s (teacher forcing) or its arg-max (attention_decoder.py:107-110):
        # step s's log-softmax kernel writes it to idx_all[s + 1] according to the DEVICE flag of that step; the GRU kernel of
        # step s + 1 reads it there, and the backward scatters through idx_all
        assert flags.dtype == torch.int32 and flags.is_cuda and flags.numel() >= S
        fused = FUSED_STEP and N <= 32 and C <= 256
        # round 6: the whole forward loop as ONE persistent launch (csrc/decode_persist.hip) that leaves the same saved buffers behind
        # as the per-step launches below.  With arg-max feedback (flags[s] == 0) the kernel scores the output layer of the previous
        # step itself -- only the arg-max is needed inside the loop -- so the log-softmax / NLL of all steps is batched behind the
        # loop in both modes
        persist = (fused and PERSIST and dtype == torch.bfloat16 and Ep % 8 == 0 and BATCHED_OUT
                   and bool(load().mr_decode_persist_ok(dt, N, T, Hd, Ep)))
        batched_out = fused and (all_teacher or persist)
        if batched_out and S > 1:
            # the word fed to step s + 1 is the target of step s.  One concatenation KERNEL: a same-dtype device-to-device
            # `copy_` is a memcpy NODE in the captured step, and a kernel that waits for a memcpy node waits tens of microseconds
            # (the 23 + 40 + 56 us holes of profiles/r06_fpn_attention_step_sequence.txt)
            idx_all = torch.cat((torch.full((1, N), int(blank), dtype=torch.int64, device=dev), targets_t[:S - 1]), 0)
        else:
            idx_all = torch.empty((S, N), dtype=torch.int64, device=dev)
            idx_all[0].fill_(int(blank))
        if persist:
            ws, ws_size = _persist_workspace(N, dev)
            call("mr_decode_persist_fwd", ptr(cat.w_n), ptr(cat.bias_d), ptr(ic.w_n), Ep, ptr(G), ldG, ptr(idx_all),
                 0 if all_teacher else ptr(flags), ptr(out.w_n), ptr(out.bias_d), C, ptr(eproj), ptr(enc), ptr(vf), ptr(H_all),
                 ptr(HC_all), ptr(W_att), ptr(CTX_all), ptr(SAVE_all), ptr(ws), ws_size, S, N, T, Ep)
        for s in range(0 if persist else S):
            call("mr_gemm_nt", dt, ptr(H_all[s]), Hd, ptr(cat.w_n), Hd, ptr(HC_all[s]), HC, ptr(cat.bias_d), 0, N, HC, Hd)
            call("mr_attn_fwd2", dt, ptr(HC_all[s]), HC, ptr(eproj), ptr(vf), ptr(enc), ptr(W_att[s]), ptr(CTX_all[s]), N, T,
                 Hd, Ep)
            if fused:
                # round 4: [GEMM(context) + GRU gates] and [output layer + log-softmax + NLL + feedback word]: 4 launches a step
                call("mr_gemm_gru_fwd", dt, ptr(CTX_all[s]), Ep, ptr(ic.w_n), Ep, ptr(G), ldG, ptr(idx_all[s]),
                     ptr(HC_all[s]) + Hd * es, HC, ptr(H_all[s]), ptr(H_all[s + 1]), ptr(SAVE_all[s]), N, Hd, Ep)
                if batched_out:
                    continue
                last = s + 1 == S
                call("mr_out_nll_fwd", dt, ptr(H_all[s + 1]), Hd, ptr(out.w_n), Hd, ptr(out.bias_d), ptr(targets_t[s]), 1,
                     ptr(mask_all[s]), ptr(LP_all[s]), ptr(loss), ptr(am_all[s]), 0 if last else ptr(flags) + 4 * s,
                     0 if last else ptr(idx_all[s + 1]), N, C, Hd, 1 if s else 0)
                continue
            call("mr_gemm_nt", dt, ptr(CTX_all[s]), Ep, ptr(ic.w_n), Ep, ptr(gic), H3, 0, 0, N, H3, Ep)
            call("mr_gru_fwd2", dt, ptr(G), ldG, ptr(idx_all[s]), ptr(gic), ptr(HC_all[s]) + Hd * es, HC, ptr(H_all[s]),
                 ptr(H_all[s + 1]), ptr(SAVE_all[s]), N, Hd)
            call("mr_gemm_nt", dt, ptr(H_all[s + 1]), Hd, ptr(out.w_n), Hd, ptr(logits), out.np_, ptr(out.bias_d), 0, N, C,
                 Hd)
            if s + 1 < S:
                call("mr_nll_step_feed_fwd", dt, ptr(logits), out.np_, ptr(targets_t[s]), 1, ptr(mask_all[s]),
                     ptr(LP_all[s]), ptr(loss), ptr(am_all[s]), ptr(flags) + 4 * s, ptr(idx_all[s + 1]), N, C, 1 if s else 0)
            else:
                call("mr_nll_step_fwd", dt, ptr(logits), out.np_, ptr(targets_t[s]), 1, ptr(mask_all[s]), ptr(LP_all[s]),
                     ptr(loss), ptr(am_all[s]), N, C, 1 if s else 0, 0)
        if batched_out:
            logits_all = torch.empty((S * N, out.np_), dtype=dtype, device=dev)     # (pad columns are never read)
            call("mr_gemm_nt", dt, ptr(H_all[1]), Hd, ptr(out.w_n), Hd, ptr(logits_all), out.np_, ptr(out.bias_d), 0, S * N, C,
                 Hd)
            loss_rows = torch.empty((S, N), dtype=torch.float32, device=dev)
            call("mr_nll_step_fwd", dt, ptr(logits_all), out.np_, ptr(targets_t), 1, ptr(mask_all), ptr(LP_all), ptr(loss_rows),
                 ptr(am_all), S * N, C, 0, 0)
            loss = loss_rows.sum(0)
        ctx.save_for_backward(G, eproj, enc, vf, H_all, HC_all, W_att, CTX_all, SAVE_all, LP_all, idx_all, mask_all,
                              targets_t)
        ctx.lin = (cat, ic, out)
        ctx.meta = (dtype, C, N, T, Hd, Ep, S, w_ic.shape[1])
        att = W_att.permute(1, 0, 2).contiguous()
        return loss, att

    @staticmethod
    def backward(ctx, gloss, gatt):
        (G, eproj, enc, vf, H_all, HC_all, W_att, CTX_all, SAVE_all, LP_all, idx_all, mask_all,
         targets_t) = ctx.saved_tensors
        cat, ic, out = ctx.lin
        dtype, C, N, T, Hd, Ep, S, E = ctx.meta
        dt = dtype_code(dtype)
        es = 2 if dtype == torch.bfloat16 else 4
        dev = enc.device
        H3, HC = 3 * Hd, 4 * Hd
        gl = gloss.to(torch.float32).contiguous()
        ga = None
        if gatt is not None:
            ga = gatt.to(torch.float32).contiguous()          # [N, S, T]
        DL_all = torch.zeros((S, N, out.np_), dtype=dtype, device=dev)
        DGI_all = torch.empty((S, N, H3), dtype=dtype, device=dev)
        DHC_all = torch.empty((S, N, HC), dtype=dtype, device=dev)
        DCTX_all = torch.empty((S, N, Ep), dtype=dtype, device=dev)
        deproj = torch.zeros((N, T, Hd), dtype=torch.float32, device=dev)
        dv = torch.zeros((Hd,), dtype=torch.float32, device=dev)
        dh_a = torch.empty((N, Hd), dtype=dtype, device=dev)     # from the next step's stacked projection
        dh_b = torch.empty((N, Hd), dtype=dtype, device=dev)     # from the next step's z * h path
        # the output layer's gradient has no recurrence in it (the arg-max feedback is detached, attention_decoder.py:110): the
        # log-softmax / NLL gradient of ALL steps is one launch over S*N rows and dh_c of all steps ONE [S*N, C] x [C, H] GEMM
        # in front of the loop, instead of two launch-latency-sized kernels per step on the backward chain (62 launches less)
        gl_all = gl.unsqueeze(0).expand(S, N).contiguous()
        call("mr_nll_step_bwd", dt, ptr(gl_all), ptr(LP_all), ptr(targets_t), 1, ptr(mask_all), ptr(DL_all), out.np_, S * N,
             C)
        DHO_all = torch.empty((S, N, Hd), dtype=dtype, device=dev)
        call("mr_gemm_nt", dt, ptr(DL_all), out.np_, ptr(out.w_t), out.np_, ptr(DHO_all), Hd, 0, 0, S * N, Hd, out.np_)
        fused = FUSED_STEP and N <= 32
        # round 6: the whole reverse loop as ONE persistent launch (csrc/decode_persist.hip); nothing in it depends on how the
        # forward chose the fed words, so it serves teacher forcing and arg-max feedback alike
        persist = (fused and PERSIST_BWD and dtype == torch.bfloat16 and Ep % 8 == 0 and ic.np_ == H3 and cat.np_ == HC
                   and bool(load().mr_decode_persist_bwd_ok(dt, N, T, Hd, Ep)))
        if persist:
            ws, ws_size = _persist_workspace(N, dev, backward=True)
            denc = torch.empty((N, T, Ep), dtype=dtype, device=dev)       # summed over the steps inside the kernel
            call("mr_decode_persist_bwd", ptr(cat.w_t), ptr(ic.w_t), H3, ptr(eproj), ptr(enc), ptr(vf), ptr(H_all), ptr(HC_all),
                 ptr(W_att), ptr(SAVE_all), ptr(DHO_all), ptr(ga) if ga is not None else 0, S * T, ptr(DGI_all), ptr(DHC_all),
                 ptr(DCTX_all), ptr(deproj), ptr(dv), ptr(denc), ptr(ws), ws_size, S, N, T, Ep)
        for s in range(-1 if persist else S - 1, -1, -1):
            last = s == S - 1
            if fused and not last:
                # round 4: the GEMM that sends the next step's stacked-projection gradient back to h' carries this step's
                # GRU backward in its epilogue (dh_a is never stored): 3 launches a step
                call("mr_gemm_gru_bwd", dt, ptr(DHC_all[s + 1]), HC, ptr(cat.w_t), HC, ptr(dh_b), ptr(DHO_all[s]),
                     ptr(SAVE_all[s]), ptr(HC_all[s]) + Hd * es, HC, ptr(H_all[s]), ptr(DGI_all[s]),
                     ptr(DHC_all[s]) + Hd * es, HC, ptr(dh_b), N, Hd, HC)
            else:
                call("mr_gru_bwd2", dt, 0 if last else ptr(dh_a), 0 if last else ptr(dh_b), ptr(DHO_all[s]),
                     ptr(SAVE_all[s]), ptr(HC_all[s]) + Hd * es, HC, ptr(H_all[s]), ptr(DGI_all[s]),
                     ptr(DHC_all[s]) + Hd * es, HC, ptr(dh_b), N, Hd)
            call("mr_gemm_nt", dt, ptr(DGI_all[s]), H3, ptr(ic.w_t), H3, ptr(DCTX_all[s]), Ep, 0, 0, N, Ep, H3)
            call("mr_attn_bwd2", dt, ptr(DCTX_all[s]), (ptr(ga) + s * T * 4) if ga is not None else 0, S * T,
                 ptr(HC_all[s]), HC, ptr(eproj), ptr(vf), ptr(enc), ptr(W_att[s]), ptr(DHC_all[s]), HC, ptr(deproj),
                 ptr(dv), N, T, Hd, Ep)
            if s > 0 and not fused:
                call("mr_gemm_nt", dt, ptr(DHC_all[s]), HC, ptr(cat.w_t), HC, ptr(dh_a), Hd, 0, 0, N, Hd, HC)
        P = S * N
        dWcat = torch.zeros((HC, Hd), dtype=torch.float32, device=dev)
        dbcat = torch.zeros((HC,), dtype=torch.float32, device=dev)
        call("mr_gemm_tn", dt, ptr(DHC_all), HC, ptr(H_all), Hd, ptr(dWcat), Hd, P, HC, Hd, 0, ptr(dbcat))
        dWic = torch.zeros((H3, Ep), dtype=torch.float32, device=dev)
        call("mr_gemm_tn", dt, ptr(DGI_all), H3, ptr(CTX_all), Ep, ptr(dWic), Ep, P, H3, Ep, 0, 0)
        dWout = torch.zeros((out.np_, Hd), dtype=torch.float32, device=dev)
        dbout = torch.zeros((out.np_,), dtype=torch.float32, device=dev)
        call("mr_gemm_tn", dt, ptr(DL_all), out.np_, ptr(H_all[1]), Hd, ptr(dWout), Hd, P, out.np_, Hd, 0, ptr(dbout))
        dG = torch.zeros((G.shape[0], G.shape[1]), dtype=torch.float32, device=dev)
        call("mr_rows_scatter_add", dt, ptr(idx_all), ptr(DGI_all), H3, ptr(dG), P, G.shape[0], G.shape[1])
        if not persist:
            denc = torch.empty((N, T, Ep), dtype=dtype, device=dev)
            call("mr_attn_denc", dt, ptr(W_att), ptr(DCTX_all), ptr(denc), S, N, T, Ep)
        return (dG.to(dtype), deproj.to(dtype), denc, dv, dWcat[:Hd], dWcat[Hd:], dbcat[Hd:], dWic[:, :E], dWout[:C],
                dbout[:C], None, None, None, None)


class Attn(nn.Module):
    """parameter holder with the reference's names / init (attention_decoder.py:134-144)."""

    def __init__(self, method, hidden_dims, embed_size):
        super(Attn, self).__init__()
        self.method = method
        self.hidden_dims = hidden_dims
        self.embed_size = embed_size
        self.attn = nn.Linear(2 * self.hidden_dims + embed_size, hidden_dims)
        self.v = nn.Parameter(torch.rand(hidden_dims))
        stdv = 1. / np.sqrt(self.v.size(0))
        self.v.data.normal_(mean=0, std=stdv)


class AttentionRNNCell(nn.Module):
    """parameter holder (attention_decoder.py:180-198); the step itself is driven by AttentionDecoder."""

    def __init__(self, hidden_dims, embedded_dims, nr_classes, n_layers=1, dropout_p=0, bidirectional=False):
        super(AttentionRNNCell, self).__init__()
        self.hidden_dims = hidden_dims
        self.embedded_dims = embedded_dims
        self.nr_classes = nr_classes
        self.n_layers = n_layers
        self.dropout_p = dropout_p
        self.embedding = nn.Embedding(nr_classes, nr_classes)
        self.embedding.weight.data = torch.eye(nr_classes)
        self.dropout = nn.Dropout(dropout_p)
        self.word_linear = nn.Linear(nr_classes, hidden_dims)
        self.attn = Attn('concat', hidden_dims, embedded_dims)
        self.rnn = nn.GRUCell(2 * hidden_dims + embedded_dims, hidden_dims)
        self.out = nn.Linear(hidden_dims, nr_classes)


class AttentionDecoder(nn.Module):
    def __init__(self, in_channels, charset=DefaultCharset(), inner_channels=512, max_size=32, height=1,
                 gt_as_output=None, step_dropout=0, **kwargs):
        super(AttentionDecoder, self).__init__()
        if step_dropout:
            raise NotImplementedError("step_dropout > 0 is not used by any reference experiment")
        self.inner_channels = inner_channels
        self.encode = self._init_encoder(in_channels)
        self.max_size = max_size
        self.charset = charset
        self.height = height
        self.decoder = AttentionRNNCell(inner_channels, max_size + height, len(charset))
        self.step_dropout = step_dropout
        self.onehot_embedding_x = nn.Embedding(max_size, max_size)
        self.onehot_embedding_x.weight.data = torch.eye(max_size)
        self.onehot_embedding_y = nn.Embedding(height, height)
        self.onehot_embedding_y.weight.data = torch.eye(height)
        self.gt_as_output = gt_as_output
        self.loss_function = nn.NLLLoss(reduction='none')

    def _init_encoder(self, in_channels, stride=(2, 1), padding=(0, 1)):
        c = self.inner_channels
        enc = nn.Sequential(
            self.conv_bn_relu(in_channels, c), self.conv_bn_relu(c, c), MaxPool2d((2, 2), (2, 2), (0, 0)),
            self.conv_bn_relu(c, c), self.conv_bn_relu(c, c), MaxPool2d(stride, stride, (0, 0)),
            self.conv_bn_relu(c, c), self.conv_bn_relu(c, c), MaxPool2d(stride, stride, (0, 0)),
            self.conv_bn_relu(c, c, kernel_size=(2, 3), stride=stride, padding=padding))
        for i in (1, 4, 7):     # the second convolution of each pair is the only consumer of the first one's BatchNorm
            enc[i][0].sole_consumer_of_bn = True
        return enc

    def _get_gt_as_output(self):
        if self.gt_as_output is not None:
            return self.gt_as_output
        return np.random.rand() < 0.5

    def _teacher_forcing_flags(self, S, dev):
        """Device int32 [S]: 1 = the target is fed to the next step, 0 = the arg-max (attention_decoder.py:107-110).  Fixed
        `gt_as_output`: a cached constant.  Otherwise one coin per step: drawn with np.random in the reference's order and
        uploaded in an eager step; inside a hipGraph capture they come from torch's device generator instead (graph-safe
        Philox state: every REPLAY draws new coins -- a host coin would be frozen into the captured step)."""
        if self.gt_as_output is not None:
            key = (bool(self.gt_as_output), S, dev)
            cache = self.__dict__.setdefault("_flag_cache", {})
            if key not in cache:
                cache[key] = torch.full((S,), int(bool(self.gt_as_output)), dtype=torch.int32, device=dev)
            return cache[key]
        if torch.cuda.is_current_stream_capturing():
            return (torch.rand((S,), device=dev) < 0.5).to(torch.int32)
        coins = [int(bool(self._get_gt_as_output())) for _ in range(S)]
        return torch.tensor(coins, dtype=torch.int32).to(dev)

    def conv_bn_relu(self, input_channels, output_channels, kernel_size=3, stride=1, padding=1):
        return nn.Sequential(Conv2d(input_channels, output_channels, kernel_size=kernel_size, stride=stride,
                                    padding=padding),
                             BatchNorm2d(output_channels, fuse_relu=True), FusedReLU())

    def _sequence(self, feature):
        """encoder features + one-hot position embeddings as [N, T, Ep] (Ep = 545 padded to one vector)."""
        dtype = get_compute_dtype()
        seq = self.encode(feature)                       # logical [N, C, height, max_size]
        N, C, Hh, Ww = seq.shape
        if Hh != self.height or Ww != self.max_size:
            raise RuntimeError("attention encoder output %dx%d does not match height=%d, max_size=%d"
                               % (Hh, Ww, self.height, self.max_size))
        dev = seq.device
        iy, ix = torch.meshgrid(torch.arange(self.height, device=dev), torch.arange(self.max_size, device=dev),
                                indexing='ij')
        emb_x = self.onehot_embedding_x(ix).to(dtype)     # [h, w, max_size]
        emb_y = self.onehot_embedding_y(iy).to(dtype)     # [h, w, height]
        T = self.height * self.max_size
        feat = seq.permute(0, 2, 3, 1).reshape(N, T, C)   # NHWC view -> [N, T, C]
        parts = [feat.to(dtype), emb_y.reshape(1, T, -1).expand(N, -1, -1), emb_x.reshape(1, T, -1).expand(N, -1, -1)]
        E = C + self.height + self.max_size
        Ep = _ceil_to(E, vec_of(dtype))
        if Ep != E:
            parts.append(torch.zeros((N, T, Ep - E), dtype=dtype, device=dev))
        return torch.cat(parts, dim=2).contiguous(), E, dtype

    def forward(self, feature, targets=None, lengths=None, train=False):
        if not feature.is_cuda:
            raise NotImplementedError("megreader_amd decoders run on the GPU only")
        enc, E, dtype = self._sequence(feature)
        N, T, Ep = enc.shape
        Hd = self.inner_channels
        cell = self.decoder
        C = len(self.charset)
        dev = enc.device
        Wa = cell.attn.attn.weight
        lin_e = _SeqLinear(Wa[:, Hd:Hd + E], cell.attn.attn.bias, Ep, dtype)
        lin_iw = _SeqLinear(cell.rnn.weight_ih[:, :Hd], cell.rnn.bias_ih, Hd, dtype)
        Cp = _ceil_to(C, vec_of(dtype))
        lin_word = _SeqLinear(cell.word_linear.weight, cell.word_linear.bias, Cp, dtype)
        eproj = lin_e(enc.view(N * T, Ep))[:, :Hd].reshape(N, T, Hd)
        if not eproj.is_contiguous():
            eproj = eproj.contiguous()
        if self.training:
            targets = targets.to(device=dev, dtype=torch.long)
            lengths_d = lengths.to(dev)
            S = self.max_size
            # the word path depends on the class index only: table G[c] = W_ih[:, :H] (word_linear(embedding[c])) + b_ih
            rows = _EmbedRowsFn.apply(torch.arange(C, device=dev), cell.embedding.weight, Cp, dtype)
            G = lin_iw(lin_word(rows))
            # teacher-forcing coins of the S steps as a DEVICE tensor (see _teacher_forcing_flags)
            flags = self._teacher_forcing_flags(S, dev)
            targets_t = targets[:, :S].t().contiguous()
            loss, att = _DecodeLoopFn.apply(G, eproj, enc, cell.attn.v, Wa[:, :Hd], cell.rnn.weight_hh, cell.rnn.bias_hh,
                                            cell.rnn.weight_ih[:, Hd:Hd + E], cell.out.weight, cell.out.bias, targets_t,
                                            lengths_d, flags, (dtype, C, int(self.charset.blank),
                                                               self.gt_as_output is not None and bool(self.gt_as_output)))
            return loss, att.view(N, -1, self.height, self.max_size)

        # ---- eval: greedy decode, one step at a time (arg-max feedback, early stop when every sample emitted a blank)
        lin_h = _SeqLinear(Wa[:, :Hd], None, Hd, dtype)
        lin_ic = _SeqLinear(cell.rnn.weight_ih[:, Hd:Hd + E], None, Ep, dtype)
        lin_hh = _SeqLinear(cell.rnn.weight_hh, cell.rnn.bias_hh, Hd, dtype)
        lin_out = _SeqLinear(cell.out.weight, cell.out.bias, Hd, dtype)
        att_state = {'dtype': dtype}
        hidden = torch.zeros((N, Hd), dtype=dtype, device=dev)
        timestep_input = torch.full((N,), int(self.charset.blank), dtype=torch.int64, device=dev)

        def step(t, word_idx, hidden):
            first = t == 0
            # embedding is a trainable [V, V] table initialised to the identity (attention_decoder.py:190-191): look
            # the row up (tiny gather, torch) and run word_linear on it
            word = lin_word(_EmbedRowsFn.apply(word_idx, cell.embedding.weight, Cp, dtype))
            w, context = _AttnStepFn.apply(lin_h(hidden), eproj, enc, cell.attn.v, att_state, first)
            hnew = _GruGatesFn.apply(lin_iw(word), lin_ic(context), lin_hh(hidden), hidden, dtype)
            return lin_out(hnew), hnew, w

        pred = torch.full((N, self.max_size), int(self.charset.blank), dtype=torch.int32, device=dev)
        probs = torch.empty((N, C), dtype=torch.float32, device=dev)
        am = torch.empty((N,), dtype=torch.int64, device=dev)
        with torch.no_grad():
            for timestep in range(self.max_size):
                logits, hidden, w = step(timestep, timestep_input, hidden)
                call("mr_nll_step_fwd", dtype_code(dtype), ptr(logits), logits.stride(0), 0, 0, 0, ptr(probs), 0,
                     ptr(am), N, C, 0, 1)
                timestep_input = am.clone()
                pred[:, timestep] = am
                if bool((am == self.charset.blank).all()):
                    break
        return pred
