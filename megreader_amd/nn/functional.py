"""torch.autograd Functions that route the training hot path through libmegreader_hip.so.

Tensor convention between layers: *logical* NCHW shape with NHWC (channels_last) strides, stored in the
compute dtype (bf16 by default, f32 in parity mode).  Weights stay fp32 ``nn.Parameter``s with the
reference's names and shapes; every forward converts them to the operand images the kernels want
(KRSC / transposed / gate-interleaved) with a small HIP kernel.

There is no CPU path in this file: CPU tensors raise NotImplementedError (same as the reference's
``ops/ctc_2d/ctc_loss_2d.py:12-13`` does for its CUDA-only op).
"""
import contextlib
import ctypes
import os

import torch
from torch.autograd import Function

from .. import get_compute_dtype
from . import prep
from .._lib import call, dtype_code, load, ptr, require_cuda, vec_of

_ESIZE = {torch.float32: 4, torch.bfloat16: 2}


def _ceil_to(x, m):
    return (x + m - 1) // m * m


def grad_sink(param, physical_shape=None):
    """Gradient sink protocol.  megreader_amd.optim attaches to every parameter it owns a persistent f32 view of its
    flat gradient buffer (`param._mr_grad_sink`, also installed as `param.grad`).  Backward kernels that accumulate
    (`+=`) anyway -- the split-P wgrad GEMMs and column sums -- write straight into that view and the autograd
    Function returns None for the parameter: no temporary gradient, no zero-fill, no `grad += tmp` kernel.
    Returns the sink tensor if it exists and its memory is dense in the layout the kernel writes
    (`physical_shape`: e.g. KRSC for a channels_last OIHW conv weight), else None (normal autograd path)."""
    sink = getattr(param, "_mr_grad_sink", None)
    if sink is None or sink.dtype != torch.float32 or not sink.is_cuda:
        return None
    # only while the sink IS param.grad: after `model.zero_grad()` (grad=None) or a user-assigned .grad the normal
    # autograd accumulation path must run, otherwise stale sums would survive in the flat buffer
    if param.grad is None or param.grad.data_ptr() != sink.data_ptr():
        return None
    if physical_shape == "strided":   # the kernel takes explicit strides
        return sink
    if physical_shape is not None:
        if len(physical_shape) == 4:
            phys = sink.permute(0, 2, 3, 1)
            if tuple(phys.shape) != tuple(physical_shape) or not phys.is_contiguous():
                return None
        elif tuple(sink.shape) != tuple(physical_shape) or not sink.is_contiguous():
            return None
    elif not sink.is_contiguous():
        return None
    return sink


def notify_grad_ready(param):
    """Tell data-parallel wrappers that `param`'s gradient was accumulated through its sink (the autograd
    post-accumulate hook does not fire when a Function returns None)."""
    for hook in getattr(param, "_mr_grad_ready_hooks", ()):
        hook(param)


class ZeroArena(object):
    """Pre-zeroed f64 scratch for the per-channel reduction buffers of BatchNorm (mr_bn_scratch_doubles(C) per call,
    forward and backward).  Each call used to zero its own buffer with a memset (120 tiny launches per ResNet-50 step).  Slices
    are handed out once per zeroing by a bump pointer; `reset()` -- called by the fused optimizers' `zero_grad()` --
    re-zeroes the used prefix with ONE fill and rewinds.  When the arena is exhausted (nobody calls reset: eval
    loops, foreign optimizers) `take` returns None and the caller falls back to its own memset."""
    SIZE = 1 << 24  # doubles (128 MB: the offset / mask gradients of the 13 DCN layers of a detector step take ~32 MB; ResNet-50 takes ~34 doubles per BN channel per step, ~1 M; the four persistent-LSTM
                    # exchange rings of a CRNN step 3.2 M)
    arenas = {}

    def __init__(self, device):
        self.buf = torch.zeros((ZeroArena.SIZE,), dtype=torch.float64, device=device)
        self.offset = 0

    @staticmethod
    def take(device, n):
        a = ZeroArena.arenas.get(device)
        if a is None:
            a = ZeroArena.arenas[device] = ZeroArena(device)
        n = (n + 31) // 32 * 32
        if a.offset + n > ZeroArena.SIZE:
            return None
        out = a.buf[a.offset:a.offset + n]
        a.offset += n
        return out

    @staticmethod
    def reset(device):
        a = ZeroArena.arenas.get(device)
        if a is not None and a.offset:
            a.buf[:a.offset].zero_()
            a.offset = 0

    @staticmethod
    def rewind(device):
        """(data pointer, bytes) of the used prefix -- which the CALLER zeroes (zero_tensors: one launch together with the flat
        gradient buffers) -- and rewind; None when nothing was handed out."""
        a = ZeroArena.arenas.get(device)
        if a is None or not a.offset:
            return None
        used = a.offset
        a.offset = 0
        return a.buf.data_ptr(), used * 8


def zero_segments(segments):
    """Zero fill of (data pointer, bytes) pairs, 8 per launch (mr_zero_multi); 16-byte aligned, sizes multiples of 16."""
    for i in range(0, len(segments), 8):
        chunk = segments[i:i + 8]
        n = len(chunk)
        call("mr_zero_multi", n, (ctypes.c_void_p * n)(*[p for p, _ in chunk]), (ctypes.c_longlong * n)(*[b for _, b in chunk]))


class _Side(object):
    """Side HIP stream for weight-gradient GEMMs that nothing later in the backward pass depends on.  The LSTM
    recurrences are launch-latency-bound chains that leave most CUs idle; the weight-gradient GEMMs of one layer run
    beside the recurrence chain of the next.  Fork = the side stream waits for the producing stream; join = an
    autograd-engine callback at the end of the backward pass makes the backward stream wait for the side stream, so
    everything after `backward()` (optimizer, gradient clipping, the next zero_grad) is ordered as usual.  Inside a
    hipGraph capture this records a fork/join in the graph."""
    stream = None
    pending = []    # tensors the side stream still reads / writes: kept alive until the join
    queued = False
    enabled = os.environ.get("MEGREADER_OVERLAP", "0") == "1"  # measured slower on the CRNN step: off

    @staticmethod
    def fork():
        main = torch.cuda.current_stream()
        if _Side.stream is None or _Side.stream.device != main.device:
            _Side.stream = torch.cuda.Stream(device=main.device)
        _Side.stream.wait_stream(main)
        if not _Side.queued:
            torch.autograd.Variable._execution_engine.queue_callback(_Side.join)
            _Side.queued = True
        return _Side.stream

    @staticmethod
    def join():
        if _Side.stream is not None:
            torch.cuda.current_stream().wait_stream(_Side.stream)
        del _Side.pending[:]
        _Side.queued = False


class _Fan(object):
    """Fork/join of INDEPENDENT launches inside one autograd Function: the weight-gradient GEMMs of an LSTM / Linear
    layer (dense TN GEMMs with 16..64 output tiles: each alone leaves most CUs with one latency-bound workgroup or none,
    tools/microbench_tn_dense.py) run on side HIP streams beside each other and beside the layer's input-gradient GEMM,
    and are joined before the Function returns -- nothing outlives the call, unlike _Side.  Inside a hipGraph capture the
    event waits become parallel branches of the graph.
    MEASURED SLOWER on the CRNN step (3.31 ms with, 3.17 ms without; gpurun_out r2p): the four GEMMs compete for the
    same CUs and lose their XCD-local tile maps; the per-launch fix that did pay is the split model in
    gemm_conv.hip:launch_tn.  Off unless MEGREADER_FAN=1."""
    streams = {}
    enabled = os.environ.get("MEGREADER_FAN", "0") == "1"

    @staticmethod
    def run(fns):
        """fns[0] runs on the current stream, fns[1:] on side streams; returns after the current stream waits for all."""
        fns = [f for f in fns if f is not None]
        if not fns:
            return
        if not _Fan.enabled or len(fns) == 1:
            for f in fns:
                f()
            return
        main = torch.cuda.current_stream()
        pool = _Fan.streams.setdefault(main.device, [])
        while len(pool) < len(fns) - 1:
            pool.append(torch.cuda.Stream(device=main.device))
        for st, f in zip(pool, fns[1:]):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                f()
        fns[0]()
        for st in pool[:len(fns) - 1]:
            main.wait_stream(st)


def accumulate_multi(pairs):
    """dst += src for a list of (dst, src) dense f32 tensor pairs, 8 pairs per launch (mr_accumulate_multi)."""
    for i in range(0, len(pairs), 8):
        chunk = pairs[i:i + 8]
        n = len(chunk)
        dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in chunk])
        src = (ctypes.c_void_p * n)(*[s.data_ptr() for _, s in chunk])
        cnt = (ctypes.c_longlong * n)(*[s.numel() for _, s in chunk])
        call("mr_accumulate_multi", n, dst, src, cnt)


class _TnDefer(object):
    """Deferred, grouped weight-gradient launches (include/megreader_hip.h: mr_tn_defer / mr_tn_flush).  Nothing later in a
    backward pass reads a weight gradient that goes to a gradient sink, so the small weight-gradient GEMMs (ResNet 1x1 / strided
    layers, LSTM / Linear layers) are recorded while backward runs and launched several problems per launch: at the latest from
    an autograd-engine callback at the end of the backward pass, earlier whenever MAX problems are waiting.  The operands of the
    recorded problems are kept alive here until the flush; parameters are reported to data-parallel wrappers
    (notify_grad_ready) only after the launch that completes their gradient.
    MEGREADER_TN_DEFER=0 (or mr_tuning.tn_defer = 0): every launch immediate (A/B)."""
    enabled = os.environ.get("MEGREADER_TN_DEFER", "1") != "0"
    MAX = 12
    # MEGREADER_TN_SIDE: where a flush in the MIDDLE of a backward pass is launched.  0: on the backward stream.  1: the group a
    # BiLSTM layer closes (its four weight gradients + the Linear layers recorded behind it) goes to the side stream (_Side),
    # beside the next layer's recurrence -- a latency chain on half of the CUs.  2: also the convolution problems, SIDE_MAX at a
    # time, beside the input-gradient chain (pays when that chain's launches leave CUs idle: small per-GPU batches).
    # Never for parameters a data-parallel wrapper watches (it wants the gradients as they complete, on the backward stream).
    # MEASURED SLOWER in every replayed step (profiles/r05_ab_tn_side_stream.txt: CRNN 2.542 -> 2.565 / 2.560 ms, 32 crops 1.154 ->
    # 1.173 / 1.210, Res50-PPM 11.21 -> 11.69, FPN-attention 8.25 -> 9.37, DB 9.05 -> 9.83 for modes 1 / 2): a grouped launch
    # holds 198 VGPRs per lane, so the recurrence's workgroups cannot share a SIMD with it and wait for its CUs anyway, and
    # beside the convolution chain the two streams evict each other's L2 lines.  Default 0; kept as an A/B switch.
    side = int(os.environ.get("MEGREADER_TN_SIDE", "0"))
    SIDE_MAX = int(os.environ.get("MEGREADER_TN_SIDE_MAX", "3"))
    keep = []
    notify = []
    queued = False
    task = -1      # graph task (one backward() call) the end-of-backward callback is registered with
    mark = 0

    @staticmethod
    def _task_id():
        fn = getattr(torch._C, "_current_graph_task_id", None)
        return fn() if fn is not None else -1

    @staticmethod
    def abort():
        """Forget everything recorded and not yet launched: the backward pass that recorded it raised -- the autograd
        engine skips its final callbacks then, so `queued` would stay set, the records would point at operands about to be
        released and a later step would add stale problems into its gradients.  Called from the recording sites when a call
        between begin() and end() raises, from begin() when it finds the state of ANOTHER graph task, and from the optimizers'
        zero_grad() / the capture fallbacks of the graphed step."""
        lib = load()
        lib.mr_tn_defer(0)
        lib.mr_tn_discard()
        del _TnDefer.keep[:]
        del _TnDefer.notify[:]
        _TnDefer.queued = False
        _TnDefer.task = -1

    @staticmethod
    def begin():
        """Start recording for the calls that follow (until end()); False when deferral is not possible here (switched off,
        or not inside a backward pass -- the end-of-backward flush could not be registered)."""
        if not _TnDefer.enabled:
            return False
        tid = _TnDefer._task_id()
        if _TnDefer.queued and tid != _TnDefer.task:
            _TnDefer.abort()          # left over from a backward pass that never reached its callback
        if not _TnDefer.queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_TnDefer.flush)
            except RuntimeError:
                return False
            _TnDefer.queued = True
            _TnDefer.task = tid
        lib = load()
        _TnDefer.mark = lib.mr_tn_pending()
        lib.mr_tn_defer(1)
        return True

    @staticmethod
    def record(launch, tensors, params):
        """begin() was True: run `launch` (the library calls to record) and end(); a raise in between drops the recording state
        instead of leaving mr_tn_defer(1) on for whatever this thread launches next."""
        try:
            launch()
        except BaseException:
            _TnDefer.abort()
            raise
        return _TnDefer.end(tensors, params)

    @staticmethod
    def end(tensors, params):
        """Stop recording.  True when the calls since begin() were recorded (the caller must then NOT report its parameters
        ready: the flush does); False when the library launched them at once (f32, all-taps kernel, mr_tuning.tn_defer = 0)."""
        lib = load()
        lib.mr_tn_defer(0)
        n = lib.mr_tn_pending()
        if n == _TnDefer.mark:
            return False
        _TnDefer.keep.extend(t for t in tensors if t is not None)
        _TnDefer.notify.extend(params)
        if n >= _TnDefer.MAX:
            _TnDefer.flush(final=False, side=_TnDefer.side >= 2)
        elif _TnDefer.side >= 2 and n >= _TnDefer.SIDE_MAX:
            _TnDefer.flush(final=False, side=True, only_side=True)
        return True

    @staticmethod
    def flush(final=True, side=False, only_side=False):
        """side: the caller is in the middle of a backward pass and nothing before its end reads these gradients -- launch on
        the side stream if MEGREADER_TN_SIDE allows (only_side: or not at all yet)."""
        if load().mr_tn_pending():
            watched = any(getattr(p, "_mr_grad_ready_hooks", None) for p in _TnDefer.notify)
            if side and not final and _TnDefer.side >= 1 and not watched:
                with torch.cuda.stream(_Side.fork()):
                    call("mr_tn_flush_beside")
                _Side.pending.extend(_TnDefer.keep)      # operands stay alive until the join at the end of the backward pass
                del _TnDefer.keep[:]
                del _TnDefer.notify[:]                   # nobody watches these parameters
                return
            if only_side:
                return
            call("mr_tn_flush")
        ready, _TnDefer.notify = _TnDefer.notify, []
        del _TnDefer.keep[:]
        if final:
            _TnDefer.queued = False
            _TnDefer.task = -1
        for p in ready:
            notify_grad_ready(p)


def discard_deferred_wgrads():
    """Drop every recorded, not yet launched weight-gradient problem (a backward pass that raised; a failed capture)."""
    _TnDefer.abort()


def flush_deferred_wgrads():
    """Launch every recorded weight-gradient problem now (data-parallel wrappers call this before they finalise a backward
    pass; harmless when nothing is recorded) and make the current stream wait for the ones already launched on the side stream."""
    _TnDefer.flush(final=False)
    if _Side.queued:
        _Side.join()


def to_internal(x, dtype):
    """logical [N,C,H,W] (any layout) -> NHWC-contiguous [N,H,W,Cp] tensor of `dtype` (view when possible)."""
    N, C, H, W = x.shape
    v = vec_of(dtype)
    xp = x.permute(0, 2, 3, 1)
    if x.dtype == dtype and xp.is_contiguous() and C % v == 0:
        return xp
    if xp.is_contiguous() and C % v == 0 and x.dtype in _ESIZE:
        out = torch.empty(xp.shape, dtype=dtype, device=x.device)
        call("mr_cast", dtype_code(x.dtype), ptr(xp), dtype_code(dtype), ptr(out), out.numel())
        return out
    xs = x.detach()
    if xs.dtype != torch.float32:
        xs = xs.float()
    xs = xs.contiguous()
    Cp = _ceil_to(C, v)
    out = torch.empty((N, H, W, Cp), dtype=dtype, device=x.device)
    call("mr_nchw_to_nhwc", dtype_code(dtype), ptr(xs), ptr(out), N, C, H, W, Cp)
    return out


def _grad_internal(g, dtype):
    """incoming gradient (logical NCHW) -> NHWC contiguous tensor of the compute dtype."""
    gp = g.permute(0, 2, 3, 1)
    if g.dtype == dtype and gp.is_contiguous():
        return gp
    return to_internal(g, dtype)


def mark_zero_padded(t):
    """Tag an NHWC gradient buffer whose channels beyond the logical count are zero (written that way by the producing
    kernel).  A consumer that works on channel-padded operands -- Conv2dFn.backward with Kp != K -- recognises a logical
    NCHW view `t[..., :K].permute(0, 3, 1, 2)` of a tagged buffer and uses the buffer itself instead of re-padding."""
    t._mr_zero_padded = True
    return t


def _zero_padded_base(gy, shape, dtype):
    """The tagged NHWC buffer behind the logical-NCHW gradient `gy`, if `gy` is exactly its [..., :K] view; else None."""
    base = gy._base
    if base is None or not getattr(base, "_mr_zero_padded", False):
        return None
    N, Ho, Wo, Kp = shape
    if base.dtype != dtype or tuple(base.shape) != (N, Ho, Wo, Kp) or not base.is_contiguous():
        return None
    if gy.data_ptr() != base.data_ptr() or gy.shape[0] != N or tuple(gy.shape[2:]) != (Ho, Wo) or \
            tuple(gy.stride()) != (Ho * Wo * Kp, 1, Wo * Kp, Kp):
        return None
    return base


def _zero_padded_rows(gy, M, Nout, Np, dtype):
    """The tagged dense [..., Np] buffer behind the gradient `gy`, if `gy` is exactly its [..., :Nout] view; else None."""
    base = gy._base
    if base is None or not getattr(base, "_mr_zero_padded", False):
        return None
    if base.dtype != dtype or not base.is_contiguous() or base.shape[-1] != Np or base.numel() != M * Np:
        return None
    if gy.data_ptr() != base.data_ptr() or gy.shape[-1] != Nout or tuple(gy.shape[:-1]) != tuple(base.shape[:-1]) or \
            tuple(gy.stride()) != tuple(base.stride()):
        return None
    return base.view(M, Np)


def _conv_out(size, k, s, p, d):
    return (size + 2 * p - d * (k - 1) - 1) // s + 1


_ROWTABS = {}


def _wgrad_rowtab(device, geom, vector_k=True):
    """Cached row table of the wgrad gather for one conv geometry (see mr_conv2d_wgrad_tab).  Returns
    (tensor, build_flag): build_flag is 1 only for the call that has to fill it.  vector_k: the output channel count is
    a whole number of vectors (the all-taps kernel is then eligible, and its table has another format)."""
    key = (device, geom, bool(vector_k))
    tab = _ROWTABS.get(key)
    if tab is not None:
        return tab, 0
    n, ho, wo = geom[0], geom[-2], geom[-1]
    if len(_ROWTABS) > 256:
        _ROWTABS.clear()
    tab = _ROWTABS[key] = torch.empty((n * ho * wo, 2), dtype=torch.int32, device=device)
    return tab, 1


from .._lib import ensure_tn_workspace as ensure_tn_taps_workspace  # noqa: E402,F401  (kept under its first name)


def set_tn_taps(mode):
    """Switch the all-taps wgrad kernel (csrc/tn_taps.hip) on / off.  The cached row tables are dropped: the two kernels
    read different table formats.  Returns the previous setting."""
    from .._lib import load
    old = load().mr_set_tn_taps(int(mode))
    _ROWTABS.clear()
    return old


def _conv_operands(weight, bias, dtype, Cp, Kp, need_dx):
    """Compute-dtype images of a conv weight: KRSC (forward / wgrad layout), CRSK (dgrad) and the padded bias.
    Persistent buffers, regenerated only when the parameter changes (megreader_amd.nn.prep)."""
    K, C, R, S = weight.shape
    dev = weight.device

    def build(old):
        if old is None:
            alloc = torch.zeros if Kp != K else torch.empty
            w_krsc = alloc((Kp, R, S, Cp), dtype=dtype, device=dev)
            w_crsk = alloc((C, R, S, Kp), dtype=dtype, device=dev) if need_dx else None
            bias_p = torch.zeros((Kp,), dtype=torch.float32, device=dev) if (bias is not None and Kp != K) else None
        else:
            w_krsc, w_crsk, bias_p = old
        jobs = [prep.conv_job(ptr(weight), weight.stride(), ptr(w_krsc), ptr(w_crsk), K, C, R, S, Cp, Kp)]
        if bias_p is not None:
            jobs.append(prep.bias_job(ptr(bias), 0, ptr(bias_p), K, 0))
        return (w_krsc, w_crsk, bias_p), jobs

    w_krsc, w_crsk, bias_p = prep.prepared((weight, bias), ("conv", Cp, Kp, bool(need_dx)), build, dtype)
    return w_krsc, w_crsk, (bias_p if bias_p is not None else bias)


# --------------------------------------------------------------------------------------------------
# Conv2d (+bias, +fused ReLU).  reference: nn.Conv2d at backbones/crnn.py:48, backbones/resnet.py:39-56
# --------------------------------------------------------------------------------------------------
class Conv2dFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, relu, relu_grad_downstream=False, bn_sums=None,
                fork=False, bnb_link=None):
        require_cuda(x, weight, bias)
        ctx.bnb_link = bnb_link
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        v = vec_of(dtype)
        xi = to_internal(x, dtype)
        N, H, W, Cp = xi.shape
        K, C, R, S = weight.shape
        if C > Cp or weight.dtype != torch.float32:
            raise RuntimeError("conv weight %s does not match input channels %d" % (tuple(weight.shape), Cp))
        Kp = _ceil_to(K, v)  # output channels padded to one 16-byte vector (pad channels: zero weights, zero bias)
        sh, sw = stride
        ph, pw = padding
        dh, dw = dilation
        Ho, Wo = _conv_out(H, R, sh, ph, dh), _conv_out(W, S, sw, pw, dw)
        need_dx = ctx.needs_input_grad[0]
        if need_dx and Cp != C:
            raise RuntimeError("input gradient requested for a channel-padded convolution input")
        w_krsc, w_crsk, bias_k = _conv_operands(weight, bias, dtype, Cp, Kp, need_dx)
        y = torch.empty((N, Ho, Wo, Kp), dtype=dtype, device=x.device)
        if bn_sums is not None and not relu and Kp == K:
            # the BatchNorm that consumes y gets its batch statistics from this GEMM's epilogue (see conv2d(bn_stats=True))
            call("mr_conv2d_fwd_stats", dt, ptr(xi), ptr(w_krsc), ptr(bias_k), ptr(y), ptr(bn_sums), N, H, W, Cp, Cp, Kp,
                 R, S, sh, sw, ph, pw, dh, dw, Ho, Wo)
        else:
            call("mr_conv2d_fwd", dt, ptr(xi), ptr(w_krsc), ptr(bias_k), ptr(y), int(relu), N, H, W, Cp, Cp, Kp, Kp, R,
                 S, sh, sw, ph, pw, dh, dw, Ho, Wo)
        ctx.save_for_backward(xi, w_crsk, y if (relu and not relu_grad_downstream) else None)
        ctx.params = (weight, bias)
        ctx.geom = (N, H, W, Cp, C, K, Kp, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo)
        ctx.relu = relu and not relu_grad_downstream  # else the consumer (max-pool) applies the ReLU mask
        ctx.has_bias = bias is not None
        ctx.dtype = dtype
        out = (y if Kp == K else y[..., :K]).permute(0, 3, 1, 2)
        if fork:
            # second output: x itself, as an output of THIS node (conv2d_fork).  Its gradient -- the residual branch of a
            # ResNet block -- arrives in backward together with gy and is added in the dgrad epilogue
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, gy, g_fork=None):
        xi, w_crsk, y = ctx.saved_tensors
        N, H, W, Cp, C, K, Kp, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo = ctx.geom
        dtype = ctx.dtype
        dt = dtype_code(dtype)
        if Kp == K:
            g = _grad_internal(gy, dtype)
        else:  # re-pad the gradient to the padded channel count (small head convolutions only)
            g = _zero_padded_base(gy, (N, Ho, Wo, Kp), dtype)      # already padded by its producer (DCN offset convs)
            if g is None:
                g = torch.zeros((N, Ho, Wo, Kp), dtype=dtype, device=gy.device)
                g[..., :K] = gy.permute(0, 2, 3, 1)
        if ctx.relu:
            gm = torch.empty_like(g)
            call("mr_relu_bwd", dt, ptr(g), ptr(y), ptr(gm), g.numel())
            g = gm
        dx = dwt = db = None
        if ctx.needs_input_grad[0]:
            dxi = torch.empty((N, H, W, C), dtype=dtype, device=g.device)
            ga = None
            if g_fork is not None:     # dx = dgrad(g) + gradient of the forked alias of x, one kernel (mr_conv2d_dgrad_add)
                ga = _grad_internal(g_fork, dtype)
                if tuple(ga.shape) != (N, H, W, C) or not ga.is_contiguous():
                    ga = ga.contiguous()
            link = ctx.bnb_link
            fused_bnb = False
            if link is not None and link.xi is not None and link.dtype == dtype and tuple(link.xi.shape) == (N, H, W, C) and \
                    BNB_EPILOGUE:
                # x is the output of a training-mode BatchNorm and this convolution its only consumer: dxi IS that
                # BatchNorm's incoming gradient, and its two per-channel reductions ride in this dgrad's epilogue
                sums = ZeroArena.take(g.device, load().mr_bn_scratch_doubles(C))
                if sums is not None:
                    produced = ctypes.c_int(0)
                    call("mr_conv2d_dgrad_bnb", dt, ptr(g), ptr(w_crsk), ptr(dxi), ptr(ga), ptr(link.xi), ptr(link.y),
                         ptr(link.mean), ptr(link.rstd), ptr(sums), ctypes.byref(produced), N, H, W, C, C, Kp, Kp, R, S, sh,
                         sw, ph, pw, dh, dw, Ho, Wo)
                    fused_bnb = True
                    if produced.value:
                        link.sums, link.grad_ptr = sums, dxi.data_ptr()
            if fused_bnb:
                pass
            elif ga is not None:
                call("mr_conv2d_dgrad_add", dt, ptr(g), ptr(w_crsk), ptr(dxi), ptr(ga), N, H, W, C, C, Kp, Kp, R, S, sh, sw,
                     ph, pw, dh, dw, Ho, Wo)
            elif POINTWISE_STRIDED_DGRAD and R == 1 and S == 1 and (sh > 1 or sw > 1) and ph == 0 and pw == 0:
                # strided 1x1 convolution (ResNet downsample branch): the dense dgrad on the sub-sampled grid, then one pass
                # that places it at the sampled positions of dx and zeroes the rest (mr_scatter_strided)
                dxs = torch.empty((N, Ho, Wo, C), dtype=dtype, device=g.device)
                call("mr_conv2d_dgrad", dt, ptr(g), ptr(w_crsk), ptr(dxs), N, Ho, Wo, C, C, Kp, Kp, 1, 1, 1, 1, 0, 0, 1, 1, Ho,
                     Wo)
                call("mr_scatter_strided", dt, ptr(dxs), ptr(dxi), N, H, W, C, sh, sw, Ho, Wo)
            else:
                call("mr_conv2d_dgrad", dt, ptr(g), ptr(w_crsk), ptr(dxi), N, H, W, C, C, Kp, Kp, R, S, sh, sw, ph, pw,
                     dh, dw, Ho, Wo)
            dx = dxi.permute(0, 3, 1, 2)
        elif g_fork is not None:
            raise RuntimeError("conv2d_fork: the forked input received a gradient but the convolution input needs none")
        weight_p, bias_p = ctx.params
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        # gradient sinks (see grad_sink): the wgrad / bias kernels accumulate directly into the flat gradient buffer
        # (padded output channels, Kp != K: the wgrad kernel writes exactly K rows / bias entries -- the pad channels of g
        # are zero -- so the sinks work for the 27-channel DCN offset convolutions too)
        w_sink = grad_sink(weight_p, (K, R, S, C)) if (ctx.needs_input_grad[1] and Cp == C) else None
        b_sink = grad_sink(bias_p, (K,)) if want_db else None
        Kw = K if (w_sink is not None or b_sink is not None) else Kp   # rows of dw / entries of db the kernels write
        if want_db:
            db = b_sink if b_sink is not None else torch.zeros((Kp,), dtype=torch.float32, device=g.device)
        if ctx.needs_input_grad[1]:
            gw = w_sink if w_sink is not None else torch.zeros((Kp, R, S, Cp), dtype=torch.float32, device=g.device)
            # the bias gradient (column sums of dy) rides along the wgrad pass over dy
            # the gather's row table (8 bytes per output pixel) depends only on the layer geometry: built on first
            # use, then passed to every later step (bf16, R*S <= 32; otherwise the call is plain mr_conv2d_wgrad)
            tab, build = (None, 0)
            if dtype == torch.bfloat16 and R * S <= 32:
                # (Kw % 8: the all-taps kernel needs whole channel vectors and its table has another format than the GEMM
                # kernel's -- two layers of one geometry that differ in that must not share a table)
                tab, build = _wgrad_rowtab(g.device, (N, H, W, Cp, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo), Kw % 8 == 0)
            # sunk gradients have no reader inside the backward pass: the launch may wait and share a grouped launch (_TnDefer)
            defer = w_sink is not None and (not want_db or b_sink is not None) and _TnDefer.begin()

            def wgrad_launch():
                call("mr_conv2d_wgrad_tab", dt, ptr(g), ptr(xi), ptr(gw), ptr(db) if want_db else 0, N, H, W, Cp, Cp, Kw,
                     Kp, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo, ptr(tab), build)
            if defer:
                if _TnDefer.record(wgrad_launch, (g, xi, tab), [weight_p] + ([bias_p] if want_db else [])):
                    return dx, None, None, None, None, None, None, None, None, None, None
            else:
                wgrad_launch()
            if w_sink is not None:
                dwt = None
                notify_grad_ready(weight_p)
            else:
                if Cp != C or Kp != K:
                    gw = gw[:K, :, :, :C]
                dwt = gw.permute(0, 3, 1, 2)
        elif want_db:
            call("mr_colsum", dt, ptr(g), ptr(db), N * Ho * Wo, Kw, Kp, 0)
        if want_db:
            if b_sink is not None:
                db = None
                notify_grad_ready(bias_p)
            elif Kp != K:
                db = db[:K]
        return dx, dwt, db, None, None, None, None, None, None, None, None


# data gradient of a strided 1x1 convolution as dense dgrad + scatter (round 4); MEGREADER_POINTWISE_STRIDED_DGRAD=0: the
# implicit-GEMM dgrad over every output pixel (A/B)
POINTWISE_STRIDED_DGRAD = os.environ.get("MEGREADER_POINTWISE_STRIDED_DGRAD", "1") != "0"

# BatchNorm-backward sums in the dgrad epilogue of the consuming convolution: OFF by default.  Measured in the step (round 4,
# gpurun r4r, same box, 40 graph replays each, on / off): CRNN 2.922 / 2.838 ms, Res50-PPM 13.955 / 12.698, FPN-attention
# 10.192 / 9.865, DB 11.449 / 10.929 -- the epilogue's element-wise reads of x and y in the MFMA accumulator layout plus its
# per-column f64 atomics, on the 4-wave tiles only, cost more than the g re-read the reduction pass saves (DESIGN.md section 4,
# "Round 4").  MEGREADER_BNB_EPILOGUE=1 turns it on; tests/test_kernels_gpu.py keeps its parity covered.
BNB_EPILOGUE = os.environ.get("MEGREADER_BNB_EPILOGUE", "0") == "1"


_DIST_GUARD = {"done": False}


def _no_resident_grids_beside_collectives():
    """The one-pass BatchNorm backward needs its whole grid resident at once (slab barrier).  Beside a collective kernel of
    ANOTHER stream -- any data-parallel run, whoever issues the collectives: the apex shim, torch's own DistributedDataParallel,
    SyncBatchNorm with user code -- workgroups could wait for CUs a collective holds, for as long as a slow peer takes.  Once
    torch.distributed is initialised with more than one rank the two-launch backward is used (not only when the
    shim is constructed).  Checked at every BatchNorm backward until a process group exists; one attribute lookup afterwards."""
    if _DIST_GUARD["done"]:
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        from ..runtime import no_resident_grid_kernels_beside_collectives
        no_resident_grid_kernels_beside_collectives(dist.get_world_size())
        _DIST_GUARD["done"] = True


class BnBwdLink(object):
    """Backward hand-over between a training-mode BatchNorm and the ONE convolution that consumes its output: the tensors the
    BatchNorm's backward reductions need (filled by BatchNormFn.forward), and -- once that convolution's backward has run --
    the f64 scratch in which its dgrad epilogue accumulated them (mr_conv2d_dgrad_bnb) plus the address of the gradient they
    describe.  BatchNormFn.backward uses the sums only if it is handed exactly that gradient tensor."""
    __slots__ = ("xi", "y", "mean", "rstd", "dtype", "sums", "grad_ptr")

    def __init__(self):
        self.xi = self.y = self.mean = self.rstd = self.dtype = self.sums = self.grad_ptr = None


class BnStatsHandoff(object):
    """Batch statistics of a conv output that its GEMM epilogue already accumulated (mr_conv2d_fwd_stats): the f64 scratch
    `sums` (a ZeroArena slice, layout of mr_bn_fwd_train) plus what identifies the tensor they describe."""
    __slots__ = ("sums", "data_ptr", "dtype", "P", "C")

    def __init__(self, sums, data_ptr, dtype, P, C):
        self.sums, self.data_ptr, self.dtype, self.P, self.C = sums, data_ptr, dtype, P, C


def conv2d(x, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), relu=False,
           relu_grad_downstream=False, bn_stats=False, fork=False, sole_consumer_of_bn=False):
    """relu_grad_downstream=True: the only consumer is a max_pool2d(..., relu_input=True), whose backward applies
    this layer's ReLU mask (saves one pass over the largest activation gradients).
    bn_stats=True: a training-mode BatchNorm consumes the output: its per-channel sum / sum of squares are accumulated in
    the epilogue of the convolution's GEMM and travel with the returned tensor (`_mr_bn_sums`); batch_norm() then skips its
    statistics pass.  Ignored (plain convolution) with a fused ReLU, padded output channels or an exhausted ZeroArena.
    fork=True: returns (y, x'), x' = x as a second output of the same autograd node.  Use x' wherever else the block needs x
    (the identity shortcut of a ResNet block, reference backbones/resnet.py:152-181): the gradients of both uses then meet in
    this convolution's backward, where the shortcut's gradient is added in the dgrad epilogue (mr_conv2d_dgrad_add) instead of
    by a separate elementwise kernel.
    sole_consumer_of_bn=True: the CALLER guarantees that x, if it is the output of a training-mode batch_norm(), is consumed by
    nothing else (the conv -> bn -> relu -> conv chains of a ResNet block).  The input gradient this convolution computes is then
    that BatchNorm's complete incoming gradient, and the BatchNorm's two backward reductions are accumulated in this dgrad's
    epilogue (mr_conv2d_dgrad_bnb) -- its backward skips the reduction pass over dy / x / y."""
    sums = None
    if bn_stats and not relu and x.is_cuda:
        K = weight.shape[0]
        if K % vec_of(get_compute_dtype()) == 0:
            sums = ZeroArena.take(x.device, load().mr_bn_scratch_doubles(K))
    want_fork = bool(fork)
    fork = want_fork and x.requires_grad and torch.is_grad_enabled()     # nothing to fuse when x needs no gradient
    link = getattr(x, "_mr_bnb_link", None) if (sole_consumer_of_bn and x.requires_grad and torch.is_grad_enabled()) else None
    y = Conv2dFn.apply(x, weight, bias, tuple(stride), tuple(padding), tuple(dilation), bool(relu),
                       bool(relu_grad_downstream), sums, fork, link)
    x_fork = x
    if fork:
        y, x_fork = y
    if sums is not None:
        y._mr_bn_sums = BnStatsHandoff(sums, y.data_ptr(), get_compute_dtype(), y.shape[0] * y.shape[2] * y.shape[3],
                                       y.shape[1])
    return (y, x_fork) if want_fork else y


# --------------------------------------------------------------------------------------------------
# BatchNorm2d (batch statistics in training).  reference: nn.BatchNorm2d at backbones/crnn.py:50
# --------------------------------------------------------------------------------------------------
class BatchNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, residual,
                num_batches_tracked=None, pre=None, link=None):
        require_cuda(x, gamma, beta)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        xi = to_internal(x, dtype)
        N, H, W, C = xi.shape
        if C != gamma.numel():
            raise RuntimeError("BatchNorm2d: channel mismatch (%d vs %d)" % (C, gamma.numel()))
        P = N * H * W
        ri = to_internal(residual, dtype) if residual is not None else None
        y = torch.empty_like(xi)
        mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        if training:
            nsum = load().mr_bn_scratch_doubles(C)   # several accumulator copies (fewer same-address atomics)
            have = (pre is not None and pre.data_ptr == xi.data_ptr() and pre.dtype == dtype and pre.P == P and
                    pre.C == C)                      # statistics already accumulated by the producing convolution
            if have:
                sums, prezeroed = pre.sums, True
            else:
                sums = ZeroArena.take(x.device, nsum)
                prezeroed = sums is not None
                if not prezeroed:
                    sums = torch.empty((nsum,), dtype=torch.float64, device=x.device)
            call("mr_bn_fwd_train", dt, ptr(xi), ptr(y), ptr(gamma), ptr(beta), ptr(running_mean),
                 ptr(running_var), ptr(mean), ptr(rstd), ptr(sums), ptr(ri),
                 int(relu) | (4 if prezeroed else 0) | (8 if have else 0), P, C, float(eps), float(momentum),
                 ptr(num_batches_tracked))
        else:
            call("mr_bn_fwd_eval", dt, ptr(xi), ptr(y), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                 ptr(mean), ptr(rstd), ptr(ri), int(relu), P, C, float(eps))
        ctx.save_for_backward(xi, y if relu else None, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.dtype = dtype
        ctx.training_mode = training
        ctx.bnb_link = None
        if link is not None and training:
            link.xi, link.y, link.mean, link.rstd, link.dtype = xi, (y if relu else None), mean, rstd, dtype
            ctx.bnb_link = link
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        if not ctx.training_mode:
            raise NotImplementedError("backward through eval-mode BatchNorm is not on the training hot path")
        _no_resident_grids_beside_collectives()
        xi, y, gamma, mean, rstd = ctx.saved_tensors
        dtype = ctx.dtype
        dt = dtype_code(dtype)
        g = _grad_internal(gy, dtype)
        N, H, W, C = xi.shape
        P = N * H * W
        dx = torch.empty_like(xi)
        dres = torch.empty_like(xi) if ctx.has_res else None
        nsum = load().mr_bn_scratch_doubles(C)
        link = ctx.bnb_link
        have = link is not None and link.sums is not None and link.grad_ptr == g.data_ptr()
        if have:      # the consuming convolution's dgrad epilogue already accumulated sum g' / sum g' xhat (BnBwdLink)
            sums, prezeroed = link.sums, True
        else:
            sums = ZeroArena.take(g.device, nsum)
            prezeroed = sums is not None
            if not prezeroed:
                sums = torch.empty((nsum,), dtype=torch.float64, device=g.device)
        if link is not None:
            link.xi = link.y = link.mean = link.rstd = link.sums = None      # one use; drop the tensor references
        gamma_p, beta_p = ctx.params
        g_sink, b_sink = grad_sink(gamma_p, (C,)), grad_sink(beta_p, (C,))
        sunk = g_sink is not None and b_sink is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        if sunk:
            dgamma, dbeta = g_sink, b_sink
        else:
            dgamma = torch.empty((C,), dtype=torch.float32, device=g.device)
            dbeta = torch.empty((C,), dtype=torch.float32, device=g.device)
        call("mr_bn_bwd", dt, ptr(g), ptr(xi), ptr(y), ptr(gamma), ptr(mean), ptr(rstd), ptr(sums), ptr(dx),
             ptr(dres), ptr(dgamma), ptr(dbeta),
             int(ctx.relu) | (2 if sunk else 0) | (4 if prezeroed else 0) | (8 if have else 0), P, C)
        gres = dres.permute(0, 3, 1, 2) if ctx.has_res else None
        if sunk:
            notify_grad_ready(gamma_p)
            notify_grad_ready(beta_p)
            dgamma = dbeta = None
        return dx.permute(0, 3, 1, 2), dgamma, dbeta, None, None, None, None, None, None, gres, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu=False, residual=None,
               num_batches_tracked=None):
    """num_batches_tracked (int64 scalar tensor, optional): incremented on the device by the statistics kernel."""
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not training):
        num_batches_tracked = None
    pre = getattr(x, "_mr_bn_sums", None) if training else None
    link = BnBwdLink() if (training and BNB_EPILOGUE and torch.is_grad_enabled() and x.requires_grad) else None
    out = BatchNormFn.apply(x, gamma, beta, running_mean, running_var, bool(training), momentum, eps, bool(relu),
                            residual, num_batches_tracked, pre, link)
    if link is not None and link.xi is not None:
        out._mr_bnb_link = link        # read by conv2d(sole_consumer_of_bn=True) of the layer that consumes `out`
    return out


# --------------------------------------------------------------------------------------------------
# Synchronised BatchNorm (reference backbones/resnet.py:26-30: apex.parallel.SyncBatchNorm when config.sync_bn; default off).
# Statistics over the batches of ALL ranks: the local per-channel sums (mr_bn_stats) are all-reduced (2C + 1 doubles), the
# normalisation runs on the same apply kernel as the eval path (mr_bn_fwd_eval with the global mean / variance).  Backward: the
# local kernels (mr_bn_bwd) give dx for LOCAL statistics; the global correction is affine in x per channel,
#     dx = dx_local + alpha_c + beta_c * x,   alpha / beta from the differences between the local and the all-reduced sums,
# so it costs one all-reduce of 2C floats and one fused elementwise pass.  dgamma / dbeta stay local sums, like apex's (the
# data-parallel wrapper averages parameter gradients afterwards).
# --------------------------------------------------------------------------------------------------
class SyncBatchNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual, num_batches_tracked,
                all_reduce):
        require_cuda(x, gamma, beta)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        xi = to_internal(x, dtype)
        N, H, W, C = xi.shape
        P = N * H * W
        dev = x.device
        ri = to_internal(residual, dtype) if residual is not None else None
        sums = torch.zeros((load().mr_bn_scratch_doubles(C),), dtype=torch.float64, device=dev)
        call("mr_bn_stats", dt, ptr(xi), ptr(sums), P, C)
        packed = torch.cat([sums[:16 * C].view(8, 2 * C).sum(dim=0), torch.full((1,), float(P), dtype=torch.float64,
                                                                                 device=dev)])
        all_reduce(packed)                                  # sum over the ranks: [sum x | sum x^2 | count]
        total = packed[2 * C]
        mean64 = packed[:C] / total
        var64 = (packed[C:2 * C] / total - mean64 * mean64).clamp_(min=0.0)
        mean_g, var_g = mean64.float(), var64.float()
        if running_mean is not None:
            with torch.no_grad():
                unbiased = var64 * (total / (total - 1.0).clamp(min=1.0))
                running_mean.mul_(1.0 - momentum).add_(mean_g, alpha=momentum)
                running_var.mul_(1.0 - momentum).add_(unbiased.float(), alpha=momentum)
                if num_batches_tracked is not None:
                    num_batches_tracked.add_(1)
        y = torch.empty_like(xi)
        mean = torch.empty((C,), dtype=torch.float32, device=dev)
        rstd = torch.empty((C,), dtype=torch.float32, device=dev)
        call("mr_bn_fwd_eval", dt, ptr(xi), ptr(y), ptr(gamma), ptr(beta), ptr(mean_g), ptr(var_g), ptr(mean), ptr(rstd),
             ptr(ri), int(relu), P, C, float(eps))
        ctx.save_for_backward(xi, y if relu else None, gamma, mean, rstd)
        ctx.total = total
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.dtype = dtype
        ctx.all_reduce = all_reduce
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xi, y, gamma, mean, rstd = ctx.saved_tensors
        dtype = ctx.dtype
        dt = dtype_code(dtype)
        g = _grad_internal(gy, dtype)
        N, H, W, C = xi.shape
        P = N * H * W
        dev = g.device
        dx = torch.empty_like(xi)
        dres = torch.empty_like(xi) if ctx.has_res else None
        sums = torch.zeros((load().mr_bn_scratch_doubles(C),), dtype=torch.float64, device=dev)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        call("mr_bn_bwd", dt, ptr(g), ptr(xi), ptr(y), ptr(gamma), ptr(mean), ptr(rstd), ptr(sums), ptr(dx), ptr(dres),
             ptr(dgamma), ptr(dbeta), int(ctx.relu) | 4, P, C)
        # dx_local = k (g' - s1/P - xhat s2/P), k = gamma rstd, s1 = dbeta, s2 = dgamma (local sums); the global version
        # replaces s/P by S/P_total: dx = dx_local + k (d1 + xhat d2) with d = s/P - S/P_total, affine in x per channel
        glob = torch.cat([dbeta, dgamma]).double()
        ctx.all_reduce(glob)
        total = ctx.total
        d1 = dbeta.double() / P - glob[:C] / total
        d2 = dgamma.double() / P - glob[C:] / total
        k = gamma.detach().double() * rstd.double()
        slope = (k * d2 * rstd.double())
        alpha = (k * d1 - slope * mean.double()).to(dtype)
        dx = torch.addcmul(dx + alpha, xi, slope.to(dtype))
        gres = dres.permute(0, 3, 1, 2) if ctx.has_res else None
        return dx.permute(0, 3, 1, 2), dgamma, dbeta, None, None, None, None, None, gres, None, None


def sync_batch_norm(x, gamma, beta, running_mean, running_var, momentum, eps, relu=False, residual=None,
                    num_batches_tracked=None, all_reduce=None):
    """Training-mode BatchNorm with statistics over all ranks.  `all_reduce(t)`: in-place SUM over the ranks of a 1-D f64
    tensor (default: torch.distributed.all_reduce on the default group)."""
    if all_reduce is None:
        import torch.distributed as dist

        def all_reduce(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return SyncBatchNormFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, bool(relu), residual,
                                 num_batches_tracked, all_reduce)


# --------------------------------------------------------------------------------------------------
# Backbone stem: Conv2d(Cin->64, 3x3, s1, p1) + ReLU + MaxPool2d(2,2) in one kernel each way.
# reference: cnn.conv0 / relu0 / pooling0 at backbones/crnn.py:17-19,48-55
# --------------------------------------------------------------------------------------------------
def stem_eligible(x, weight, stride, padding, dilation, pool_kernel, pool_stride, pool_padding):
    return (x.is_cuda and x.dim() == 4 and not x.requires_grad and x.shape[1] in (1, 3) and
            tuple(weight.shape) == (64, x.shape[1], 3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1) and
            tuple(dilation) == (1, 1) and tuple(pool_kernel) == (2, 2) and tuple(pool_stride) == (2, 2) and
            tuple(pool_padding) == (0, 0) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and
            16 * (x.shape[3] + 2) * x.shape[1] <= 48 * 1024)


class StemConvReluPoolFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        require_cuda(x, weight, bias)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        N, C, H, W = x.shape
        y = torch.empty((N, H // 2, W // 2, 64), dtype=dtype, device=x.device)
        code = torch.empty((N, H // 2, W // 2, 64), dtype=torch.uint8, device=x.device)
        sk, sc, sr, ss = weight.stride()
        wpack = None
        if dtype == torch.bfloat16:  # filter bank pre-packed for the MFMA kernel (regenerated with the other images)
            def build(old):
                buf = old[0] if old is not None else torch.empty((64, 32), dtype=dtype, device=x.device)
                return (buf,), [prep.stem_job(ptr(weight), (sk, sc, sr, ss), ptr(buf), C)]

            wpack = prep.prepared((weight,), ("stem", C), build, dtype)[0]
        call("mr_stem_fwd", dt, ptr(x), ptr(weight), sk, sc, sr, ss, ptr(wpack), ptr(bias), ptr(y), ptr(code), N, C,
             H, W)
        ctx.save_for_backward(x, code)
        ctx.params = (weight, bias)
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x, code = ctx.saved_tensors
        weight_p, bias_p = ctx.params
        dtype = ctx.dtype
        N, C, H, W = x.shape
        g = _grad_internal(gy, dtype)
        want_dw = ctx.needs_input_grad[1]
        want_db = bias_p is not None and ctx.needs_input_grad[2]
        w_sink = grad_sink(weight_p, "strided") if want_dw else None
        b_sink = grad_sink(bias_p, (64,)) if want_db else None
        dw = db = None
        if want_dw:
            dw = w_sink if w_sink is not None else torch.zeros_like(weight_p)
        if want_db:
            db = b_sink if b_sink is not None else torch.zeros((64,), dtype=torch.float32, device=g.device)
        if dw is not None or db is not None:
            ws = torch.empty((load().mr_stem_bwd_workspace(C),), dtype=torch.float32, device=g.device)
            sk, sc, sr, ss = dw.stride() if dw is not None else (0, 0, 0, 0)
            call("mr_stem_bwd", dtype_code(dtype), ptr(g), ptr(code), ptr(x), ptr(ws), ptr(dw), sk, sc, sr, ss, ptr(db),
                 N, C, H, W)
        if w_sink is not None:
            notify_grad_ready(weight_p)
            dw = None
        if b_sink is not None:
            notify_grad_ready(bias_p)
            db = None
        return None, dw, db


def stem_conv_relu_pool(x, weight, bias):
    return StemConvReluPoolFn.apply(x, weight, bias)


# --------------------------------------------------------------------------------------------------
# MaxPool2d.  reference: nn.MaxPool2d at backbones/crnn.py:17-31
# --------------------------------------------------------------------------------------------------
class ConvReluPoolFn(Function):
    """conv + bias + ReLU + max-pool as ONE forward launch (mr_conv2d_fwd_pool; reference backbones/crnn.py:14-33: Conv2d -> ReLU ->
    MaxPool2d): the pooled activation and the arg-max codes come straight out of the GEMM's epilogue, the full-resolution
    activation is never written.  Backward = the two nodes it replaces, unchanged: mr_maxpool_bwd (which also applies the ReLU
    mask, read at pooled resolution) expands the gradient, then Conv2dFn's backward runs on it."""

    @staticmethod
    def forward(ctx, x, weight, bias, padding, pool_kernel, pool_stride, pool_padding):
        require_cuda(x, weight, bias)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        xi = to_internal(x, dtype)
        N, H, W, Cp = xi.shape
        K, C, R, S = weight.shape
        ph, pw = padding
        Ho, Wo = _conv_out(H, R, 1, ph, 1), _conv_out(W, S, 1, pw, 1)
        kh, kw = pool_kernel
        psh, psw = pool_stride
        pph, ppw = pool_padding
        PHo, PWo = (Ho + 2 * pph - kh) // psh + 1, (Wo + 2 * ppw - kw) // psw + 1
        need_dx = ctx.needs_input_grad[0]
        w_krsc, w_crsk, bias_k = _conv_operands(weight, bias, dtype, Cp, K, need_dx)
        y = torch.empty((N, PHo, PWo, K), dtype=dtype, device=x.device)
        idx = torch.empty((N, PHo, PWo, K), dtype=torch.uint8, device=x.device)
        call("mr_conv2d_fwd_pool", dt, ptr(xi), ptr(w_krsc), ptr(bias_k), ptr(y), ptr(idx), 1, N, H, W, Cp, Cp, K, R, S, 1, 1, ph, pw,
             1, 1, Ho, Wo, kh, kw, psh, psw, pph, ppw, PHo, PWo)
        ctx.save_for_backward(xi, w_crsk, idx, y)
        ctx.params = (weight, bias)
        ctx.geom = (N, H, W, Cp, C, K, K, R, S, 1, 1, ph, pw, 1, 1, Ho, Wo)
        ctx.pool = (kh, kw, psh, psw, pph, ppw, PHo, PWo)
        ctx.has_bias = bias is not None
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xi, w_crsk, idx, y = ctx.saved_tensors
        N, H, W, Cp, C, K, Kp, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo = ctx.geom
        kh, kw, psh, psw, pph, ppw, PHo, PWo = ctx.pool
        dtype = ctx.dtype
        g = _grad_internal(gy, dtype)
        dz = torch.empty((N, Ho, Wo, K), dtype=dtype, device=g.device)
        call("mr_maxpool_bwd", dtype_code(dtype), ptr(g), ptr(idx), ptr(y), ptr(dz), N, Ho, Wo, K, kh, kw, psh, psw, pph, ppw,
             PHo, PWo)

        class _ConvCtx(object):      # what Conv2dFn.backward reads from its ctx (relu already applied by the pool's backward)
            pass
        c = _ConvCtx()
        c.saved_tensors = (xi, w_crsk, None)
        c.geom, c.dtype, c.params, c.has_bias = ctx.geom, dtype, ctx.params, ctx.has_bias
        c.relu, c.bnb_link = False, None
        c.needs_input_grad = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]) + (False,) * 8
        grads = Conv2dFn.backward(c, dz.permute(0, 3, 1, 2))
        return grads[0], grads[1], grads[2], None, None, None, None


def conv_relu_pool_eligible(x, weight, stride, padding, dilation, pool_kernel, pool_stride, pool_padding):
    """True when conv + ReLU + max-pool of this geometry runs as one forward launch (bf16, stride-1 convolution whose 8-wave
    tiles can be cut on pooling-window / image boundaries; see mr_conv2d_fwd_pool_ok)."""
    if not x.is_cuda or get_compute_dtype() != torch.bfloat16 or tuple(stride) != (1, 1) or tuple(dilation) != (1, 1):
        return False
    if os.environ.get("MEGREADER_CONV_POOL", "1") == "0":
        return False
    N, C, H, W = x.shape
    K, Cw, R, S = weight.shape
    if C != Cw or C % 8 != 0 or K % 8 != 0:
        return False
    ph, pw = padding
    Ho, Wo = _conv_out(H, R, 1, ph, 1), _conv_out(W, S, 1, pw, 1)
    return bool(load().mr_conv2d_fwd_pool_ok(dtype_code(torch.bfloat16), N, H, W, C, C, K, R, S, 1, 1, ph, pw, 1, 1, Ho, Wo,
                                             pool_kernel[0], pool_kernel[1], pool_stride[0], pool_stride[1],
                                             pool_padding[0], pool_padding[1]))


def conv_relu_pool(x, weight, bias, padding, pool_kernel, pool_stride, pool_padding):
    return ConvReluPoolFn.apply(x, weight, bias, tuple(padding), tuple(pool_kernel), tuple(pool_stride), tuple(pool_padding))


class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding, relu_input=False):
        require_cuda(x)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        xi = to_internal(x, dtype)
        N, H, W, C = xi.shape
        kh, kw = kernel
        sh, sw = stride
        ph, pw = padding
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        y = torch.empty((N, Ho, Wo, C), dtype=dtype, device=x.device)
        idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
        call("mr_maxpool_fwd", dt, ptr(xi), ptr(y), ptr(idx), N, H, W, C, kh, kw, sh, sw, ph, pw, Ho, Wo)
        ctx.save_for_backward(idx, y if relu_input else None)  # the ReLU mask is read at pooled resolution
        ctx.geom = (N, H, W, C, kh, kw, sh, sw, ph, pw, Ho, Wo)
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        idx, relu_y = ctx.saved_tensors
        N, H, W, C, kh, kw, sh, sw, ph, pw, Ho, Wo = ctx.geom
        dtype = ctx.dtype
        g = _grad_internal(gy, dtype)
        dx = torch.empty((N, H, W, C), dtype=dtype, device=g.device)
        call("mr_maxpool_bwd", dtype_code(dtype), ptr(g), ptr(idx), ptr(relu_y), ptr(dx), N, H, W, C, kh, kw, sh, sw,
             ph, pw, Ho, Wo)
        return dx.permute(0, 3, 1, 2), None, None, None, None


def max_pool2d(x, kernel, stride=None, padding=(0, 0), relu_input=False):
    """relu_input=True: x is the output of a ReLU whose backward mask is fused into this op's backward."""
    stride = kernel if stride is None else stride
    return MaxPoolFn.apply(x, tuple(kernel), tuple(stride), tuple(padding), bool(relu_input))


# --------------------------------------------------------------------------------------------------
# [N,C,1,W] feature map -> [W,N,C] sequence (squeeze(2).permute(2,0,1) at reference decoders/crnn.py:88-89)
# --------------------------------------------------------------------------------------------------
class MapToSequenceFn(Function):
    @staticmethod
    def forward(ctx, x):
        require_cuda(x)
        dtype = get_compute_dtype()
        xi = to_internal(x, dtype)  # [N,1,W,C]
        N, H, W, C = xi.shape
        if H != 1:
            raise AssertionError("the height of conv must be 1")
        out = torch.empty((W, N, C), dtype=dtype, device=x.device)
        call("mr_permute_021", dtype_code(dtype), ptr(xi), ptr(out), N, W, C)
        ctx.shape = (N, W, C)
        ctx.dtype = dtype
        return out

    @staticmethod
    def backward(ctx, g):
        N, W, C = ctx.shape
        dtype = ctx.dtype
        if g.dtype != dtype or not g.is_contiguous():
            g = g.to(dtype).contiguous()
        dx = torch.empty((N, 1, W, C), dtype=dtype, device=g.device)
        call("mr_permute_021", dtype_code(dtype), ptr(g), ptr(dx), W, N, C)
        return dx.permute(0, 3, 1, 2)


def map_to_sequence(x):
    return MapToSequenceFn.apply(x)


# --------------------------------------------------------------------------------------------------
# Linear.  reference: nn.Linear at decoders/crnn.py:14,21 (BidirectionalLSTM.embedding)
# Output columns are padded to a 16-byte multiple; the returned tensor is the [:, :N] view of that buffer.
# --------------------------------------------------------------------------------------------------
class LinearFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        require_cuda(x, weight, bias)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        v = vec_of(dtype)
        lead = x.shape[:-1]
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.dtype != dtype or not x2.is_contiguous():
            x2 = x2.to(dtype).contiguous()
        M = x2.shape[0]
        Nout = weight.shape[0]
        Np = _ceil_to(Nout, v)
        if weight.dim() != 2 or weight.stride() != (K, 1):
            raise RuntimeError("linear: weight must be a dense row-major [out_features, in_features] tensor "
                               "(got strides %s); call .contiguous() on views" % (tuple(weight.stride()),))
        if K % v:
            raise RuntimeError("Linear: in_features (%d) must be a multiple of %d" % (K, v))
        def build(old):
            if old is None:
                w_n_ = torch.empty((Nout, K), dtype=dtype, device=x.device)
                w_t_ = (torch.zeros if Np != Nout else torch.empty)((K, Np), dtype=dtype, device=x.device)
            else:
                w_n_, w_t_ = old
            return (w_n_, w_t_), [prep.matrix_job(ptr(weight), K, ptr(w_n_), K, ptr(w_t_), Np, Nout, K, 0)]

        w_n, w_t = prep.prepared((weight,), ("linear", Np), build, dtype)
        # (Np != Nout: the padding columns of y stay unwritten -- the only tensor handed out is the [:, :Nout] view, and the
        # kernels that take the padded buffer itself (the CTC head: logits with ldl = Np) read Nout columns)
        y = torch.empty((M, Np), dtype=dtype, device=x.device)
        call("mr_gemm_nt", dt, ptr(x2), K, ptr(w_n), K, ptr(y), Np, ptr(bias), 0, M, Nout, K)
        ctx.save_for_backward(x2, w_t)
        ctx.params = (weight, bias)
        ctx.dims = (M, K, Nout, Np)
        ctx.lead = lead
        ctx.has_bias = bias is not None
        ctx.dtype = dtype
        return y[:, :Nout].view(*lead, Nout) if Np == Nout else y.view(*lead, Np)[..., :Nout]

    @staticmethod
    def backward(ctx, gy):
        x2, w_t = ctx.saved_tensors
        M, K, Nout, Np = ctx.dims
        dtype = ctx.dtype
        dt = dtype_code(dtype)
        g2 = gy.reshape(-1, Nout) if gy.shape[-1] == Nout else gy
        if Np != Nout:
            gp = _zero_padded_rows(gy, M, Nout, Np, dtype)   # producer already wrote a zero-padded [M, Np] buffer (CTC head)
            if gp is None:
                gp = torch.zeros((M, Np), dtype=dtype, device=gy.device)
                gp[:, :Nout] = g2
        else:
            gp = g2 if (g2.dtype == dtype and g2.is_contiguous()) else g2.to(dtype).contiguous()
        dx = dw = db = None
        dx_fn = None
        if ctx.needs_input_grad[0]:
            dx2 = torch.empty((M, K), dtype=dtype, device=gy.device)
            dx_fn = lambda: call("mr_gemm_nt", dt, ptr(gp), Np, ptr(w_t), Np, ptr(dx2), K, 0, 0, M, K, Np)
            dx = dx2.view(*ctx.lead, K)
        weight_p, bias_p = ctx.params
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        # sinks also when the output width is padded (38 classes in a 40-column buffer): the GEMM is then asked for
        # NA = Nout rows of A^T B with lda = Np -- the padding columns feed output rows that are never stored
        w_sink = grad_sink(weight_p, (Nout, K)) if ctx.needs_input_grad[1] else None
        b_sink = grad_sink(bias_p, (Nout,)) if (want_db and (w_sink is not None or not ctx.needs_input_grad[1])) else None
        NA = Nout if (w_sink is not None or (not ctx.needs_input_grad[1] and b_sink is not None)) else Np
        gb = None
        if want_db:
            gb = b_sink if b_sink is not None else torch.zeros((Np,), dtype=torch.float32, device=gy.device)
        if ctx.needs_input_grad[1]:
            gw = w_sink if w_sink is not None else torch.zeros((Np, K), dtype=torch.float32, device=gy.device)
            # sunk gradients have no consumer inside the backward pass: side stream (see _Side)
            overlap = (w_sink is not None and (gb is None or b_sink is not None) and _Side.enabled and
                       not getattr(weight_p, "_mr_grad_ready_hooks", None))
            if overlap:
                if dx_fn is not None:
                    dx_fn()
                with torch.cuda.stream(_Side.fork()):
                    call("mr_gemm_tn", dt, ptr(gp), Np, ptr(x2), K, ptr(gw), K, M, NA, K, 0, ptr(gb))
                _Side.pending.extend((gp, x2))
            elif (w_sink is not None and (gb is None or b_sink is not None) and not _Fan.enabled and
                  dtype == torch.bfloat16 and _TnDefer.begin()):
                # sunk gradients: recorded for a grouped launch (_TnDefer); the flush reports the parameters ready
                def lin_launch():
                    if dx_fn is not None:
                        dx_fn()
                    call("mr_gemm_tn", dt, ptr(gp), Np, ptr(x2), K, ptr(gw), K, M, NA, K, 0, ptr(gb))
                if _TnDefer.record(lin_launch, (gp, x2), [weight_p] + ([bias_p] if want_db else [])):
                    return dx, None, None
            else:   # input-gradient and weight-gradient GEMMs are independent: side by side (_Fan)
                _Fan.run([dx_fn, lambda: call("mr_gemm_tn", dt, ptr(gp), Np, ptr(x2), K, ptr(gw), K, M, NA, K, 0,
                                              ptr(gb))])
            dx_fn = None
            if w_sink is not None:
                notify_grad_ready(weight_p)
            else:
                dw = gw[:Nout]
        elif want_db:
            call("mr_colsum", dt, ptr(gp), ptr(gb), M, NA, Np, 0)
        if dx_fn is not None:
            dx_fn()
        if want_db:
            if b_sink is not None:
                notify_grad_ready(bias_p)
            else:
                db = gb[:Nout]
        return dx, dw, db


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


# --------------------------------------------------------------------------------------------------
# nn.ConvTranspose2d(cin, cout, 2, 2) of the DB heads (reference decoders/seg_detector.py:66-79).  Kernel = stride: every output
# pixel receives exactly one input pixel, so the layer is a GEMM on the weight's own [Cin][Cout*4] matrix plus a depth-to-space
# pass (mr_deconv2x2_d2s, which also adds the bias).  One autograd node: prepared weight images (no per-step permute / repeat /
# contiguous launches), the weight gradient accumulated straight into the fused optimizer's sink -- the transposed GEMM's
# [Cin][Cout*4] result IS the parameter's layout -- and deferred into the grouped weight-gradient launch.
# --------------------------------------------------------------------------------------------------
class ConvTranspose2x2Fn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        require_cuda(x, weight, bias)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        v = vec_of(dtype)
        xi = to_internal(x, dtype)                        # [N, H, W, Cin]
        N, H, W, Cin = xi.shape
        if tuple(weight.shape[2:]) != (2, 2) or weight.shape[0] != Cin or not weight.is_contiguous():
            raise RuntimeError("conv_transpose2x2: weight must be a dense [Cin, Cout, 2, 2] tensor matching the input")
        Cout = weight.shape[1]
        K4 = 4 * Cout
        ld2 = _ceil_to(K4, v)
        P = N * H * W
        dev = x.device

        def build(old):
            if old is None:
                # w_n: the weight as [Cin][ld2] (zero padding columns), B operand of dx = dy2 . W^T; w_t: [4*Cout][Cin], B of fwd
                w_n_ = (torch.zeros if ld2 != K4 else torch.empty)((Cin, ld2), dtype=dtype, device=dev)
                w_t_ = torch.empty((K4, Cin), dtype=dtype, device=dev)
            else:
                w_n_, w_t_ = old
            return (w_n_, w_t_), [prep.matrix_job(ptr(weight), K4, ptr(w_n_), ld2, ptr(w_t_), Cin, Cin, K4, 0)]

        w_n, w_t = prep.prepared((weight,), ("deconv2x2", ld2), build, dtype)
        x2 = xi.reshape(P, Cin)
        y2 = torch.empty((P, ld2), dtype=dtype, device=dev)
        call("mr_gemm_nt", dt, ptr(x2), Cin, ptr(w_t), Cin, ptr(y2), ld2, 0, 0, P, K4, Cin)
        y = torch.empty((N, 2 * H, 2 * W, Cout), dtype=dtype, device=dev)
        call("mr_deconv2x2_d2s", dt, ptr(y2), ld2, ptr(bias), ptr(y), N, H, W, Cout)
        ctx.save_for_backward(x2, w_n)
        ctx.params = (weight, bias)
        ctx.geom = (N, H, W, Cin, Cout, ld2)
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x2, w_n = ctx.saved_tensors
        N, H, W, Cin, Cout, ld2 = ctx.geom
        dtype = ctx.dtype
        dt = dtype_code(dtype)
        P, K4 = N * H * W, 4 * Cout
        dev = gy.device
        if Cout % vec_of(dtype) == 0:
            g = _grad_internal(gy, dtype)                 # [N, 2H, 2W, Cout]
            if not g.is_contiguous():
                g = g.contiguous()
        else:   # the 1-channel maps of the heads' last layer: dense NHWC (to_internal would pad the channels to a vector)
            g = gy.permute(0, 2, 3, 1).contiguous().to(dtype)
        dy2 = torch.empty((P, ld2), dtype=dtype, device=dev)
        call("mr_deconv2x2_s2d", dt, ptr(g), ptr(dy2), ld2, N, H, W, Cout)
        weight_p, bias_p = ctx.params
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dxi = torch.empty((N, H, W, Cin), dtype=dtype, device=dev)
            call("mr_gemm_nt", dt, ptr(dy2), ld2, ptr(w_n), ld2, ptr(dxi), Cin, 0, 0, P, Cin, ld2)
            dx = dxi.permute(0, 3, 1, 2)
        want_db = bias_p is not None and ctx.needs_input_grad[2]
        if want_db:
            b_sink = grad_sink(bias_p, (Cout,))
            gb = b_sink if b_sink is not None else torch.zeros((Cout,), dtype=torch.float32, device=dev)
            call("mr_colsum", dt, ptr(g), ptr(gb), N * 4 * H * W, Cout, Cout, 0)
            if b_sink is not None:
                notify_grad_ready(bias_p)
            else:
                db = gb
        if ctx.needs_input_grad[1]:
            w_sink = grad_sink(weight_p)      # dense in the parameter's own (row-major) layout
            if ld2 != K4:     # Cout = 1: the GEMM needs whole vectors of columns -- a padded scratch, folded into the gradient
                tmp = torch.zeros((Cin, ld2), dtype=torch.float32, device=dev)
                call("mr_gemm_tn", dt, ptr(x2), Cin, ptr(dy2), ld2, ptr(tmp), ld2, P, Cin, ld2, 0, 0)
                if w_sink is not None:
                    w_sink.view(Cin, K4).add_(tmp[:, :K4])
                    notify_grad_ready(weight_p)
                else:
                    dw = tmp[:, :K4].reshape(Cin, Cout, 2, 2)
            else:
                gw = w_sink if w_sink is not None else torch.zeros((Cin, K4), dtype=torch.float32, device=dev)
                defer = w_sink is not None and dtype == torch.bfloat16 and _TnDefer.begin()

                # dW[ci, (co, i, j)] += sum_p x[p, ci] dy2[p, (co, i, j)]: the parameter's own memory order
                def deconv_launch():
                    call("mr_gemm_tn", dt, ptr(x2), Cin, ptr(dy2), ld2, ptr(gw), K4, P, Cin, K4, 0, 0)
                if not defer:
                    deconv_launch()
                if defer and _TnDefer.record(deconv_launch, (x2, dy2), [weight_p]):
                    pass
                elif w_sink is not None:
                    notify_grad_ready(weight_p)
                else:
                    dw = gw.view(Cin, Cout, 2, 2)
        return dx, dw, db


class DBHeadTailFn(Function):
    """(binary, thresh, thresh_binary) of the DB heads from the two 1-channel logit maps (reference decoders/seg_detector.py:77-79
    nn.Sigmoid at the end of `binarize` / `thresh`, :142-147 step_function): float32 outputs whatever the compute dtype, one
    launch each way (the torch expression is two casts, two sigmoids and five elementwise ops forward, ~14 launches backward)."""

    @staticmethod
    def forward(ctx, xb, xt, k):
        require_cuda(xb, xt)
        if xb.dtype != xt.dtype or xb.dtype not in _ESIZE or xb.shape != xt.shape:
            raise RuntimeError("db_head_tail: the two logit maps must share shape and dtype (float32 / bfloat16)")
        # logical [N,1,H,W]; with one channel every dense layout is the same memory order
        xbc = xb if xb.is_contiguous() or xb.permute(0, 2, 3, 1).is_contiguous() else xb.contiguous()
        xtc = xt if xt.is_contiguous() or xt.permute(0, 2, 3, 1).is_contiguous() else xt.contiguous()
        n = xb.numel()
        shape = tuple(xb.shape)
        binary = torch.empty(shape, dtype=torch.float32, device=xb.device)
        thresh = torch.empty(shape, dtype=torch.float32, device=xb.device)
        tbinary = torch.empty(shape, dtype=torch.float32, device=xb.device)
        call("mr_db_head_tail_fwd", dtype_code(xb.dtype), ptr(xbc), ptr(xtc), ptr(binary), ptr(thresh), ptr(tbinary), n, float(k))
        ctx.save_for_backward(binary, thresh, tbinary)
        ctx.meta = (xb.dtype, shape, float(k))
        ctx.set_materialize_grads(False)
        return binary, thresh, tbinary

    @staticmethod
    def backward(ctx, gb, gt, gtb):
        binary, thresh, tbinary = ctx.saved_tensors
        dtype, shape, k = ctx.meta
        gs = [None if g is None else (g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous())
              for g in (gb, gt, gtb)]
        dxb = torch.empty(shape, dtype=dtype, device=binary.device)
        dxt = torch.empty(shape, dtype=dtype, device=binary.device)
        call("mr_db_head_tail_bwd", dtype_code(dtype), ptr(binary), ptr(thresh), ptr(tbinary), ptr(gs[0]), ptr(gs[1]), ptr(gs[2]),
             ptr(dxb), ptr(dxt), binary.numel(), k)
        return dxb, dxt, None


def db_head_tail(xb, xt, k):
    return DBHeadTailFn.apply(xb, xt, k)


def conv_transpose2x2(x, weight, bias=None):
    return ConvTranspose2x2Fn.apply(x, weight, bias)


# --------------------------------------------------------------------------------------------------
# Bidirectional LSTM.  reference: nn.LSTM(nIn, nHidden, bidirectional=True) at decoders/crnn.py:13
# --------------------------------------------------------------------------------------------------
LSTM_STATUS = None   # tests set this to a list: status words of the persistent-recurrence workspaces handed out
LSTM_LOCAL = None    # ... and this one to a list of word 1 (workgroups that found their batch group on one XCD)


def _lstm_workspace(dt, T, N, H, dev):
    """(exchange buffer, size argument) of the persistent (one launch per layer and pass) recurrence, or (None, 0) when the
    library runs one launch per step for this problem (mr_lstm_ws_bytes == 0).  The buffer has to be zero at launch: it comes
    from the pre-zeroed arena (re-zeroed once per step by the fused optimizers' zero_grad; size passed NEGATIVE = "already
    zeroed") or, when the arena is exhausted, from torch's allocator (the C call then zeroes it with its own memset node)."""
    nbytes = load().mr_lstm_ws_bytes(dt, T, N, H)
    if nbytes <= 0:
        return None, 0
    size = nbytes
    arena = ZeroArena.take(dev, (nbytes + 7) // 8)
    if arena is not None:
        ws = arena.view(torch.uint8)[:nbytes]
        size = -nbytes
    else:
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    if LSTM_STATUS is not None:
        LSTM_STATUS.append(ws[nbytes - 256:nbytes - 252])
    if LSTM_LOCAL is not None:
        LSTM_LOCAL.append(ws[nbytes - 252:nbytes - 248])
    return ws, size


class BiLSTMFn(Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        require_cuda(x, w_ih)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        v = vec_of(dtype)
        es = _ESIZE[dtype]
        if x.dtype != dtype or not x.is_contiguous():
            x = x.to(dtype).contiguous()
        T, N, I = x.shape
        H = w_hh.shape[1]
        if I % v or H % v:
            raise RuntimeError("LSTM: input (%d) and hidden (%d) sizes must be multiples of %d" % (I, H, v))
        dev = x.device
        sources = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r)

        def build(old):
            if old is None:
                wcat_ = torch.empty((8 * H, I), dtype=dtype, device=dev)
                wcat_t_ = torch.empty((I, 8 * H), dtype=dtype, device=dev)
                whh_ = torch.empty((2, 4 * H, H), dtype=dtype, device=dev)
                whh_t_ = torch.empty((2, H, 4 * H), dtype=dtype, device=dev)
                bcat_ = torch.empty((8 * H,), dtype=torch.float32, device=dev)
            else:
                wcat_, wcat_t_, whh_, whh_t_, bcat_ = old
            jobs = []
            for d, (wi, wh, bi, bh) in enumerate(((w_ih, w_hh, b_ih, b_hh), (w_ih_r, w_hh_r, b_ih_r, b_hh_r))):
                # gate-major (i,f,g,o) rows -> gate-interleaved rows; both directions side by side
                jobs.append(prep.matrix_job(ptr(wi), I, ptr(wcat_) + d * 4 * H * I * es, I,
                                            ptr(wcat_t_) + d * 4 * H * es, 8 * H, 4 * H, I, H))
                jobs.append(prep.matrix_job(ptr(wh), H, ptr(whh_) + d * 4 * H * H * es, H,
                                            ptr(whh_t_) + d * 4 * H * H * es, 4 * H, 4 * H, H, H))
                jobs.append(prep.bias_job(ptr(bi), ptr(bh), ptr(bcat_) + d * 4 * H * 4, 4 * H, H))
            return (wcat_, wcat_t_, whh_, whh_t_, bcat_), jobs

        wcat, wcat_t, whh, whh_t, bcat = prep.prepared(sources, ("bilstm",), build, dtype)
        xproj = torch.empty((T * N, 8 * H), dtype=dtype, device=dev)
        call("mr_gemm_nt", dt, ptr(x), I, ptr(wcat), I, ptr(xproj), 8 * H, ptr(bcat), 0, T * N, 8 * H, I)
        out = torch.empty((T, N, 2 * H), dtype=dtype, device=dev)
        cbuf = torch.empty((T, N, 2 * H), dtype=torch.float32, device=dev)
        gates = torch.empty((T, N, 8 * H), dtype=dtype, device=dev)
        ws, ws_size = _lstm_workspace(dt, T, N, H, dev)
        call("mr_lstm_fwd", dt, ptr(xproj), ptr(whh), ptr(out), ptr(cbuf), ptr(gates), T, N, H, ptr(ws), ws_size)
        ctx.save_for_backward(x, wcat_t, whh_t, out, cbuf, gates)
        ctx.params = sources
        ctx.dims = (T, N, I, H)
        ctx.dtype = dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        x, wcat_t, whh_t, out, cbuf, gates = ctx.saved_tensors
        T, N, I, H = ctx.dims
        dtype = ctx.dtype
        dt = dtype_code(dtype)
        es = _ESIZE[dtype]
        dev = gout.device
        if gout.dtype != dtype or not gout.is_contiguous():
            gout = gout.to(dtype).contiguous()
        ws, ws_size = _lstm_workspace(dt, T, N, H, dev)
        dc = torch.empty((N, 2 * H), dtype=torch.float32, device=dev) if ws is None else None
        # NOTE: `gates` is rewritten in place with the pre-activation gradients (single backward pass only)
        call("mr_lstm_bwd", dt, ptr(gout), ptr(whh_t), ptr(cbuf), ptr(gates), ptr(dc), T, N, H, ptr(ws), ws_size)
        dgates = gates
        sinks = [grad_sink(p, tuple(p.shape)) for p in ctx.params]
        use_sinks = all(s is not None for s in sinks)
        # with sinks nothing downstream consumes the weight gradients: run their GEMMs on the side stream, beside the
        # next layer's recurrence chain (not under data-parallel hooks, which want the gradients as they complete)
        overlap = (use_sinks and _Side.enabled and
                   not any(getattr(p, "_mr_grad_ready_hooks", None) for p in ctx.params))
        side = _Side.fork() if overlap else None
        dx = dx_fn = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((T, N, I), dtype=dtype, device=dev)
            dx_fn = lambda: call("mr_gemm_nt", dt, ptr(dgates), 8 * H, ptr(wcat_t), 8 * H, ptr(dx), I, 0, 0, T * N, I,
                                 8 * H)
        if (use_sinks and not overlap and not _Fan.enabled and dtype == torch.bfloat16 and T > 1 and _TnDefer.begin()):
            # Grouped launch (_TnDefer): the four weight-gradient GEMMs of the layer -- W_ih and W_hh of both directions, each
            # straight into its sink, the bias sums (column sums of dgates, shared by b_ih and b_hh) into both bias sinks --
            # plus whatever the Linear layers behind this one recorded run as ONE launch.
            def lstm_launch():
                if dx_fn is not None:
                    dx_fn()
                P = (T - 1) * N
                for d in range(2):    # bias gradient = column sums of dgates, added to b_ih AND b_hh by the same launch
                    call("mr_gemm_tn2", dt, ptr(dgates) + d * 4 * H * es, 8 * H, ptr(x), I, ptr(sinks[4 * d]), I, T * N, 4 * H, I,
                         H, ptr(sinks[4 * d + 2]), ptr(sinks[4 * d + 3]))
                # forward direction: dgates[t] (t >= 1) with h[t-1]; reverse direction: dgates[t] (t <= T-2) with h[t+1]
                call("mr_gemm_tn", dt, ptr(dgates) + N * 8 * H * es, 8 * H, ptr(out), 2 * H, ptr(sinks[1]), H, P, 4 * H, H, H, 0)
                call("mr_gemm_tn", dt, ptr(dgates) + 4 * H * es, 8 * H, ptr(out) + (N * 2 * H + H) * es, 2 * H, ptr(sinks[5]), H,
                     P, 4 * H, H, H, 0)
            if _TnDefer.record(lstm_launch, (dgates, x, out), list(ctx.params)):
                # one launch for the layer (and the Linear layers recorded behind it); a data-parallel wrapper wants the
                # gradients as they complete, and the next layer's recurrence is a latency chain that leaves the CUs idle anyway
                # (MEGREADER_TN_SIDE >= 1: beside that chain, on the side stream)
                _TnDefer.flush(final=False, side=True)
            else:
                for p in ctx.params:
                    notify_grad_ready(p)
            return (dx,) + (None,) * 8
        if overlap and dx_fn is not None:
            dx_fn()
        with torch.cuda.stream(side) if overlap else contextlib.nullcontext():
            # one zeroed scratch: [2,4H,I] input-weight gradients (both directions from ONE GEMM) + [2,4H] bias sums
            nfl = 2 * 4 * H * (I + 1)
            # pre-zeroed arena slice (saves a fill launch) ONLY when the scratch is consumed inside this call (sinks): without
            # sinks views of it are returned as parameter gradients, and a later ZeroArena.reset() would hand the same bytes
            # to somebody else while p.grad still points at them
            arena = None if (overlap or not use_sinks) else ZeroArena.take(dev, (nfl + 1) // 2)
            scratch = (arena.view(torch.float32)[:nfl] if arena is not None else
                       torch.zeros((nfl,), dtype=torch.float32, device=dev))
            gw_ih = scratch[:2 * 4 * H * I].view(2, 4 * H, I)
            gb = scratch[2 * 4 * H * I:].view(2, 4 * H)
            if use_sinks:   # recurrent-weight gradients accumulate straight into the flat gradient buffer
                g_hh = (sinks[1], sinks[5])
            else:
                gw_hh = torch.zeros((2, 4 * H, H), dtype=torch.float32, device=dev)
                g_hh = (gw_hh[0], gw_hh[1])
            P = (T - 1) * N
            fns = [
                # bias gradient = column sums of dgates, fused into the same pass
                lambda: call("mr_gemm_tn", dt, ptr(dgates), 8 * H, ptr(x), I, ptr(gw_ih), I, T * N, 8 * H, I, H,
                             ptr(gb)),
                # forward direction: dgates[t] (t>=1) with h[t-1]
                (lambda: call("mr_gemm_tn", dt, ptr(dgates) + N * 8 * H * es, 8 * H, ptr(out), 2 * H, ptr(g_hh[0]),
                              H, P, 4 * H, H, H, 0)) if T > 1 else None,
                # reverse direction: dgates[t] (t<=T-2) with h[t+1]
                (lambda: call("mr_gemm_tn", dt, ptr(dgates) + 4 * H * es, 8 * H, ptr(out) + (N * 2 * H + H) * es,
                              2 * H, ptr(g_hh[1]), H, P, 4 * H, H, H, 0)) if T > 1 else None]
            if overlap:
                for f in fns:
                    if f is not None:
                        f()
            else:   # the four GEMMs (dx + three weight gradients) are independent: side by side (_Fan)
                _Fan.run([dx_fn] + fns)
            if use_sinks:
                # w_ih, b_ih, b_hh of both directions: one launch folds the scratch into the six sinks
                accumulate_multi([(sinks[0], gw_ih[0]), (sinks[4], gw_ih[1]), (sinks[2], gb[0]), (sinks[3], gb[0]),
                                  (sinks[6], gb[1]), (sinks[7], gb[1])])
        if overlap:
            _Side.pending.extend((dgates, x, out, scratch))
        if use_sinks:
            for p in ctx.params:
                notify_grad_ready(p)
            return (dx,) + (None,) * 8
        return (dx, gw_ih[0], g_hh[0], gb[0], gb[0].clone(), gw_ih[1], g_hh[1], gb[1], gb[1].clone())


def bilstm(x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
    return BiLSTMFn.apply(x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r)


# --------------------------------------------------------------------------------------------------
# 1-D CTC loss fused with log-softmax.  reference: decoders/crnn.py:96-98
#   pred = log_softmax(pred, dim=2).to(float64); loss = nn.CTCLoss(zero_infinity=True)(pred, targets, [T]*b, lengths)
# Returns (loss f64 scalar, log_probs f32 [T,N,C]).
# --------------------------------------------------------------------------------------------------
_FULL_LENGTHS = {}


class CTCLossFn(Function):
    @staticmethod
    def forward(ctx, logits, targets, input_lengths, target_lengths, blank, zero_infinity, want_f64=False):
        require_cuda(logits, targets, target_lengths)
        T, N, C = logits.shape
        dtype = logits.dtype
        if dtype not in _ESIZE:
            raise TypeError("ctc_loss: logits must be float32 or bfloat16")
        if logits.stride(2) != 1 or logits.stride(1) * N != logits.stride(0):
            logits = logits.contiguous()
        ldl = logits.stride(1)
        if targets.dim() != 2 or targets.shape[0] != N:
            raise RuntimeError("ctc_loss: targets must be padded [N, S]")
        S = targets.shape[1]
        targets = targets.contiguous()
        if targets.dtype not in (torch.int32, torch.int64):
            targets = targets.long()
        if input_lengths is None:   # nn.CTCLoss with input_lengths = [T] * N (decoders/crnn.py:97): a cached constant, no fill launch
            key = (N, T, logits.device)
            input_lengths = _FULL_LENGTHS.get(key)
            if input_lengths is None:
                if len(_FULL_LENGTHS) > 64:
                    _FULL_LENGTHS.clear()
                input_lengths = _FULL_LENGTHS[key] = torch.full((N,), T, dtype=torch.int64, device=logits.device)
        input_lengths = input_lengths.to(device=logits.device, dtype=torch.int64).contiguous()
        target_lengths = target_lengths.to(device=logits.device, dtype=torch.int64).contiguous()
        dev = logits.device
        lp = torch.empty((T, N, C), dtype=torch.float32, device=dev)
        # the float64 copy the reference returns as `pred` (decoders/crnn.py:96): written by the same kernel
        lp64 = torch.empty((T, N, C), dtype=torch.float64, device=dev) if want_f64 else None
        alpha = torch.empty((N, T, 2 * S + 1), dtype=torch.float64, device=dev)
        # beta is produced by the same kernel (concurrently with alpha) when a gradient may be asked for
        beta = torch.empty((N, T, 2 * S + 1), dtype=torch.float64, device=dev) if ctx.needs_input_grad[0] else None
        nll = torch.empty((N,), dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float64, device=dev)
        t64 = int(targets.dtype == torch.int64)
        call("mr_ctc_fwd", dtype_code(dtype), ptr(logits), ldl, ptr(targets), t64, ptr(input_lengths),
             ptr(target_lengths), 1, T, N, C, S, int(blank), int(zero_infinity), ptr(lp), ptr(alpha), ptr(beta),
             ptr(nll), ptr(loss), ptr(lp64))
        ctx.save_for_backward(lp, alpha, beta, nll, targets, input_lengths, target_lengths)
        ctx.dims = (T, N, C, S, int(blank), int(zero_infinity), t64)
        ctx.dtype = dtype
        out_lp = lp64 if want_f64 else lp
        ctx.mark_non_differentiable(out_lp)
        ctx.set_materialize_grads(False)     # no zero-filled [T, N, C] gradient for the log-probabilities nobody differentiates
        if int(zero_infinity) & 2:   # per-sample losses nll_b / L_b (reference decoders/ctc_loss.py:118-122)
            return nll / target_lengths.to(torch.float64), out_lp
        return loss, out_lp

    @staticmethod
    def backward(ctx, gloss, _glp):
        lp, alpha, beta, nll, targets, input_lengths, target_lengths = ctx.saved_tensors
        T, N, C, S, blank, zero_inf, t64 = ctx.dims
        dtype = ctx.dtype
        v = vec_of(dtype)
        Cp = _ceil_to(C, v)
        if gloss is None:
            return None, None, None, None, None, None, None
        g = gloss.to(torch.float64).contiguous()
        grad = torch.empty((T, N, Cp), dtype=dtype, device=lp.device)    # the kernel writes the padding columns as zeros
        call("mr_ctc_bwd", dtype_code(dtype), ptr(lp), ptr(alpha), ptr(beta), ptr(nll), ptr(targets), t64,
             ptr(input_lengths),
             ptr(target_lengths), 1, ptr(g), T, N, C, S, blank, zero_inf, ptr(grad), Cp)
        if Cp != C:
            mark_zero_padded(grad)    # the padded Linear in front takes the buffer as it is (LinearFn.backward)
        return grad[..., :C], None, None, None, None, None, None


def ctc_loss_logits(logits, targets, input_lengths, target_lengths, blank=0, zero_infinity=True, per_sample=False,
                    log_probs_f64=False):
    """CTC loss of log_softmax(logits) (fused); returns (loss, log_probs).  Default: nn.CTCLoss(reduction='mean')
    semantics (f64 scalar).  per_sample=True: the [N] vector nll_b / L_b of the reference's own python CTCLoss
    (decoders/ctc_loss.py:118-122, reduction='mean' there means "divide by the target length", no batch mean).
    log_probs_f64=True: the returned log-probabilities are float64 (the f32 values widened by the kernel itself -- the
    reference's `log_softmax(pred, dim=2).to(torch.float64)`) instead of float32."""
    flags = int(bool(zero_infinity)) | (2 if per_sample else 0)
    return CTCLossFn.apply(logits, targets, input_lengths, target_lengths, blank, flags, bool(log_probs_f64))


def softmax_eval_nc1t(logits):
    """eval head of CRNNDecoder: logits [T,N,C] -> softmax over classes as f32 [N,C,1,T]
    (reference decoders/crnn.py:101-104).  Inference only (no autograd)."""
    require_cuda(logits)
    T, N, C = logits.shape
    lg = logits.detach()
    if lg.stride(2) != 1 or lg.stride(1) * N != lg.stride(0):
        lg = lg.contiguous()
    out = torch.empty((N, C, 1, T), dtype=torch.float32, device=lg.device)
    call("mr_softmax_nc1t", dtype_code(lg.dtype), ptr(lg), lg.stride(1), ptr(out), T, N, C)
    return out


# --------------------------------------------------------------------------------------------------
# PPM / FPN helpers.  reference: backbones/ppm.py:11-44, backbones/fpn_top_down.py:16-30
# --------------------------------------------------------------------------------------------------
class AdaptiveAvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, out_hw):
        require_cuda(x)
        dtype = get_compute_dtype()
        xi = to_internal(x, dtype)
        N, H, W, C = xi.shape
        OH, OW = out_hw
        y = torch.empty((N, OH, OW, C), dtype=dtype, device=x.device)
        call("mr_adaptive_avgpool_fwd", dtype_code(dtype), ptr(xi), ptr(y), N, H, W, C, OH, OW)
        ctx.geom = (N, H, W, C, OH, OW)
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        N, H, W, C, OH, OW = ctx.geom
        g = _grad_internal(gy, ctx.dtype)
        dx = torch.empty((N, H, W, C), dtype=ctx.dtype, device=g.device)
        call("mr_adaptive_avgpool_bwd", dtype_code(ctx.dtype), ptr(g), ptr(dx), N, H, W, C, OH, OW)
        return dx.permute(0, 3, 1, 2), None


def adaptive_avg_pool2d(x, output_size):
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    return AdaptiveAvgPoolFn.apply(x, tuple(output_size))


class AdaptiveAvgPoolMultiFn(Function):
    """Several adaptive average pools of ONE map (pyramid pooling, reference backbones/ppm.py:13-20,36-40) as one autograd
    node: one launch reads the map once for all scales; backward, one launch writes the input gradient of all scales (autograd
    otherwise adds four full-size gradients of the head's largest activation)."""

    @staticmethod
    def forward(ctx, x, sizes):
        require_cuda(x)
        dtype = get_compute_dtype()
        xi = to_internal(x, dtype)
        N, H, W, C = xi.shape
        n = len(sizes)
        ys = [torch.empty((N, oh, ow, C), dtype=dtype, device=x.device) for oh, ow in sizes]
        oh = (ctypes.c_int * n)(*[s_[0] for s_ in sizes])
        ow = (ctypes.c_int * n)(*[s_[1] for s_ in sizes])
        call("mr_adaptive_avgpool_multi_fwd", dtype_code(dtype), ptr(xi), (ctypes.c_void_p * n)(*[y.data_ptr() for y in ys]),
             oh, ow, n, N, H, W, C)
        ctx.geom = (N, H, W, C, tuple(sizes))
        ctx.dtype = dtype
        return tuple(y.permute(0, 3, 1, 2) for y in ys)

    @staticmethod
    def backward(ctx, *gys):
        N, H, W, C, sizes = ctx.geom
        dtype = ctx.dtype
        n = len(sizes)
        dev = next(g for g in gys if g is not None).device
        gs = [_grad_internal(g, dtype) if g is not None else torch.zeros((N, oh, ow, C), dtype=dtype, device=dev)
              for g, (oh, ow) in zip(gys, sizes)]
        dx = torch.empty((N, H, W, C), dtype=dtype, device=dev)
        oh = (ctypes.c_int * n)(*[s_[0] for s_ in sizes])
        ow = (ctypes.c_int * n)(*[s_[1] for s_ in sizes])
        call("mr_adaptive_avgpool_multi_bwd", dtype_code(dtype), (ctypes.c_void_p * n)(*[g.data_ptr() for g in gs]), oh, ow, n,
             ptr(dx), N, H, W, C)
        return dx.permute(0, 3, 1, 2), None


def adaptive_avg_pool2d_multi(x, output_sizes):
    """[adaptive_avg_pool2d(x, s) for s in output_sizes] in one launch each way (<= 8 scales, H * W <= 512 pixels);
    falls back to the separate pools otherwise."""
    sizes = [(s_, s_) if isinstance(s_, int) else tuple(s_) for s_ in output_sizes]
    if not (1 <= len(sizes) <= 8) or x.shape[2] * x.shape[3] * 128 > 32768 or x.shape[1] % vec_of(get_compute_dtype()) \
            or sum(a * b for a, b in sizes) > 256 or max(x.shape[2], x.shape[3]) > 128:
        return tuple(adaptive_avg_pool2d(x, s_) for s_ in sizes)
    return AdaptiveAvgPoolMultiFn.apply(x, tuple(sizes))


class BilinearFn(Function):
    @staticmethod
    def forward(ctx, x, out_hw):
        require_cuda(x)
        dtype = get_compute_dtype()
        xi = to_internal(x, dtype)
        N, H, W, C = xi.shape
        OH, OW = out_hw
        y = torch.empty((N, OH, OW, C), dtype=dtype, device=x.device)
        call("mr_bilinear_fwd", dtype_code(dtype), ptr(xi), ptr(y), N, H, W, C, OH, OW, C, 0, 0)
        ctx.geom = (N, H, W, C, OH, OW)
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        N, H, W, C, OH, OW = ctx.geom
        g = _grad_internal(gy, ctx.dtype)
        dx = torch.empty((N, H, W, C), dtype=ctx.dtype, device=g.device)
        call("mr_bilinear_bwd", dtype_code(ctx.dtype), ptr(g), ptr(dx), N, H, W, C, OH, OW, C, 0)
        return dx.permute(0, 3, 1, 2), None


def interpolate_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False)."""
    return BilinearFn.apply(x, tuple(size))


class BilinearCatFn(Function):
    """torch.cat([base] + [F.interpolate(b, base.shape[2:], mode='bilinear') for b in branches], 1) -- the pyramid pooling
    head's concatenation (reference backbones/ppm.py:36-42): every resized branch is written straight into its channel slice
    of the concatenation buffer, and backward reads the slices in place (no per-branch copy either way)."""

    @staticmethod
    def forward(ctx, base, *branches):
        require_cuda(base)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        bi = to_internal(base, dtype)
        parts = [to_internal(b, dtype) for b in branches]
        N, OH, OW, C0 = bi.shape
        Ct = C0 + sum(p.shape[3] for p in parts)
        y = torch.empty((N, OH, OW, Ct), dtype=dtype, device=bi.device)
        call("mr_copy_channels", dt, ptr(bi), C0, 0, ptr(y), Ct, 0, N * OH * OW, C0)
        off = C0
        geoms = []
        for p in parts:
            _, H, W, C = p.shape
            call("mr_bilinear_fwd", dt, ptr(p), ptr(y), N, H, W, C, OH, OW, Ct, off, 0)
            geoms.append((H, W, C, off))
            off += C
        ctx.geom = (N, OH, OW, C0, Ct, tuple(geoms))
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        N, OH, OW, C0, Ct, geoms = ctx.geom
        g = _grad_internal(gy, ctx.dtype)
        dt = dtype_code(ctx.dtype)
        d0 = torch.empty((N, OH, OW, C0), dtype=ctx.dtype, device=g.device)
        call("mr_copy_channels", dt, ptr(g), Ct, 0, ptr(d0), C0, 0, N * OH * OW, C0)
        outs = [d0.permute(0, 3, 1, 2)]
        for H, W, C, off in geoms:
            d = torch.empty((N, H, W, C), dtype=ctx.dtype, device=g.device)
            call("mr_bilinear_bwd", dt, ptr(g), ptr(d), N, H, W, C, OH, OW, Ct, off)
            outs.append(d.permute(0, 3, 1, 2))
        return tuple(outs)


def cat_bilinear(base, branches):
    """cat([base] + [interpolate_bilinear(b, base.shape[2:]) for b in branches], 1); channel counts must be multiples of one
    16-byte vector (falls back to the separate ops otherwise)."""
    v = vec_of(get_compute_dtype())
    if base.shape[1] % v or any(b.shape[1] % v for b in branches):
        return cat_channels([base] + [interpolate_bilinear(b, base.shape[2:]) for b in branches])
    return BilinearCatFn.apply(base, *branches)


class UpsampleAddFn(Function):
    """F.interpolate(x, size=y.shape[2:], mode='bilinear') + y  (FPN top-down, backbones/fpn_top_down.py:16-19)."""

    @staticmethod
    def forward(ctx, x, y):
        require_cuda(x, y)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        xi, yi = to_internal(x, dtype), to_internal(y, dtype)
        N, H, W, C = xi.shape
        _, OH, OW, C2 = yi.shape
        if C != C2:
            raise RuntimeError("upsample_add: channel mismatch")
        out = torch.empty_like(yi)
        call("mr_copy_channels", dt, ptr(yi), C, 0, ptr(out), C, 0, N * OH * OW, C)
        call("mr_bilinear_fwd", dt, ptr(xi), ptr(out), N, H, W, C, OH, OW, C, 0, 1)
        ctx.geom = (N, H, W, C, OH, OW)
        ctx.dtype = dtype
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g_out):
        N, H, W, C, OH, OW = ctx.geom
        g = _grad_internal(g_out, ctx.dtype)
        dx = torch.empty((N, H, W, C), dtype=ctx.dtype, device=g.device)
        call("mr_bilinear_bwd", dtype_code(ctx.dtype), ptr(g), ptr(dx), N, H, W, C, OH, OW, C, 0)
        return dx.permute(0, 3, 1, 2), g_out


def upsample_add(x, y):
    return UpsampleAddFn.apply(x, y)


class NearestUpFn(Function):
    """nn.Upsample(scale_factor=s, mode='nearest')(x) (+ y): the DB head's top-down path and its x2 / x4 / x8 output
    branches (decoders/seg_detector.py:22-43,121-128).  With `into` = (buffer NHWC, channel offset) the result is written
    straight into a channel slice of a concatenation buffer (torch.cat((p5, p4, p3, p2), 1), seg_detector.py:130)."""

    @staticmethod
    def forward(ctx, x, scale, add):
        require_cuda(x)
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        xi = to_internal(x, dtype)
        N, H, W, C = xi.shape
        ai = to_internal(add, dtype) if add is not None else None
        out = torch.empty((N, H * scale, W * scale, C), dtype=dtype, device=xi.device)
        call("mr_nearest_up_fwd", dt, ptr(xi), ptr(ai), ptr(out), N, H, W, C, int(scale), C, 0)
        ctx.geom = (N, H, W, C, int(scale))
        ctx.dtype = dtype
        ctx.has_add = add is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g_out):
        N, H, W, C, s = ctx.geom
        g = _grad_internal(g_out, ctx.dtype)
        dx = torch.empty((N, H, W, C), dtype=ctx.dtype, device=g.device)
        call("mr_nearest_up_bwd", dtype_code(ctx.dtype), ptr(g), ptr(dx), N, H, W, C, s, C, 0)
        return dx.permute(0, 3, 1, 2), None, (g_out if ctx.has_add else None)


def upsample_nearest(x, scale, add=None):
    return NearestUpFn.apply(x, scale, add)


class CatChannelsFn(Function):
    """torch.cat(tensors, 1) for NHWC-internal tensors whose channel counts are multiples of one vector."""

    @staticmethod
    def forward(ctx, *xs):
        dtype = get_compute_dtype()
        dt = dtype_code(dtype)
        parts = [to_internal(x, dtype) for x in xs]
        N, H, W, _ = parts[0].shape
        chans = [p.shape[3] for p in parts]
        Ct = sum(chans)
        y = torch.empty((N, H, W, Ct), dtype=dtype, device=parts[0].device)
        off = 0
        for p, c in zip(parts, chans):
            call("mr_copy_channels", dt, ptr(p), c, 0, ptr(y), Ct, off, N * H * W, c)
            off += c
        ctx.chans = chans
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        g = _grad_internal(gy, ctx.dtype)
        N, H, W, Ct = g.shape
        dt = dtype_code(ctx.dtype)
        outs, off = [], 0
        for c in ctx.chans:
            d = torch.empty((N, H, W, c), dtype=ctx.dtype, device=g.device)
            call("mr_copy_channels", dt, ptr(g), Ct, off, ptr(d), c, 0, N * H * W, c)
            outs.append(d.permute(0, 3, 1, 2))
            off += c
        return tuple(outs)


def cat_channels(tensors):
    return CatChannelsFn.apply(*tensors)


class ScaleChannelsFn(Function):
    """y[n,c,h,w] = x[n,c,h,w] * scale[n,c]  (Dropout2d with a precomputed keep-mask / (1-p))."""

    @staticmethod
    def forward(ctx, x, scale):
        dtype = get_compute_dtype()
        xi = to_internal(x, dtype)
        N, H, W, C = xi.shape
        y = torch.empty_like(xi)
        call("mr_scale_channels", dtype_code(dtype), ptr(xi), ptr(scale), ptr(y), N, H * W, C)
        ctx.save_for_backward(scale)
        ctx.dtype = dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (scale,) = ctx.saved_tensors
        g = _grad_internal(gy, ctx.dtype)
        N, H, W, C = g.shape
        dx = torch.empty_like(g)
        call("mr_scale_channels", dtype_code(ctx.dtype), ptr(g), ptr(scale), ptr(dx), N, H * W, C)
        return dx.permute(0, 3, 1, 2), None


def dropout2d(x, p, training):
    """nn.Dropout2d: whole channels are zeroed per sample with probability p; kept channels scale by 1/(1-p).
    The keep-mask comes from torch's generator (bernoulli on [N, C]); it cannot match the reference's RNG stream."""
    if not training or p == 0.0:
        return x
    N, C = x.shape[0], x.shape[1]
    keep = torch.empty((N, C), dtype=torch.float32, device=x.device).bernoulli_(1.0 - p)
    return ScaleChannelsFn.apply(x, keep / (1.0 - p))


# --------------------------------------------------------------------------------------------------
# 2D-CTC head: pred = log(max(softmax_H(mask) * softmax_C(classify), tiny)).permute(3,2,0,1)
# reference: decoders/ctc_decoder2d.py:16-45
# --------------------------------------------------------------------------------------------------
class CTC2DHeadFn(Function):
    @staticmethod
    def forward(ctx, mask_logits, cls_logits, tiny):
        """mask_logits logical [N,1,H,W], cls_logits logical [N,C,H,W] (outputs of the HIP convs).
        Returns (lp [W,H,N,C] f32, mask_prob [N,1,H,W] f32, cls_prob logical [N,C,H,W] f32)."""
        require_cuda(mask_logits, cls_logits)
        a = mask_logits.permute(0, 2, 3, 1)
        z = cls_logits.permute(0, 2, 3, 1)
        if a.stride(3) != 1 or z.stride(3) != 1 or a.dtype != z.dtype:
            raise RuntimeError("ctc2d head expects channel-contiguous NHWC logits")
        N, H, W, C = z.shape
        lda, ldz = a.stride(2), z.stride(2)
        for t, ld in ((a, lda), (z, ldz)):
            if t.stride(1) != W * ld or t.stride(0) != H * W * ld:
                raise RuntimeError("ctc2d head expects dense (row-padded) NHWC logits")
        dev = z.device
        lp = torch.empty((W, H, N, C), dtype=torch.float32, device=dev)
        m = torch.empty((N, H, W), dtype=torch.float32, device=dev)
        p = torch.empty((N, H, W, C), dtype=torch.float32, device=dev)
        call("mr_ctc2d_head_fwd", dtype_code(z.dtype), ptr(a), lda, ptr(z), ldz, ptr(lp), ptr(m), ptr(p), N, H, W, C,
             float(tiny))
        ctx.save_for_backward(m, p)
        ctx.geom = (N, H, W, C, float(tiny))
        ctx.dtype = z.dtype
        ctx.mark_non_differentiable(m, p)
        return lp, m.unsqueeze(1), p.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, glp, _gm, _gp):
        m, p = ctx.saved_tensors
        N, H, W, C, tiny = ctx.geom
        dtype = ctx.dtype
        v = vec_of(dtype)
        Cp = _ceil_to(C, v)
        g = glp.to(torch.float32).contiguous()
        da = torch.zeros((N, H, W, v), dtype=dtype, device=g.device)
        dz = torch.zeros((N, H, W, Cp), dtype=dtype, device=g.device) if Cp != C else \
            torch.empty((N, H, W, Cp), dtype=dtype, device=g.device)
        call("mr_ctc2d_head_bwd", dtype_code(dtype), ptr(g), ptr(m), ptr(p), ptr(da), v, ptr(dz), Cp, N, H, W, C, tiny)
        return da[..., :1].permute(0, 3, 1, 2), dz[..., :C].permute(0, 3, 1, 2), None


def ctc2d_head(mask_logits, cls_logits, tiny):
    return CTC2DHeadFn.apply(mask_logits, cls_logits, tiny)
