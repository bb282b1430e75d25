"""Prepared operand images of fp32 master weights (compute-dtype KRSC / CRSK / gate-interleaved copies).

Every layer of the hot path consumes its weights in a kernel-friendly image (reference layers consume the fp32
parameter directly: nn.Conv2d at backbones/crnn.py:48, nn.LSTM / nn.Linear at decoders/crnn.py:13-14).  Producing
those images used to cost ~3 small kernels per layer per step.  This module keeps them in persistent buffers:

* a layer asks `prepared(params, key, build)`; on a hit the buffers are returned as they are, on a miss `build`
  (the layer's own individual `mr_prep_*` calls) fills them and the entry is recorded on the owning parameter;
* an entry is valid while every source parameter still has the same storage pointer and autograd version
  (`load_state_dict`, `copy_`, torch.optim updates all bump the version -> miss -> rebuilt);
* megreader_amd.optim updates parameters through raw pointers (no version bump), so after each update it calls
  `refresh(params)`, which regenerates ALL recorded images of those parameters with ONE `mr_prep_batch` launch
  driven by a job table kept in device memory (rebuilt only when the set of entries changes -- never inside a
  captured hipGraph step once the warm-up steps have run).

Weights mutated behind autograd's back (`p.data.add_(...)`) are not detected; call `invalidate(module)` after such
an edit.  (The reference only does this at construction time: backbones/resnet.py:218-221.)
"""
import ctypes

import torch

from .._lib import call, dtype_code, ptr

KIND_CONV, KIND_MATRIX, KIND_BIAS, KIND_STEM = 0, 1, 2, 3


class PrepJob(ctypes.Structure):
    """struct mr_prep_job (include/megreader_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("src2", ctypes.c_void_p), ("dst_a", ctypes.c_void_p),
                ("dst_b", ctypes.c_void_p), ("s0", ctypes.c_longlong), ("s1", ctypes.c_longlong),
                ("s2", ctypes.c_longlong), ("s3", ctypes.c_longlong), ("kind", ctypes.c_int), ("d0", ctypes.c_int),
                ("d1", ctypes.c_int), ("d2", ctypes.c_int), ("d3", ctypes.c_int), ("pad", ctypes.c_int),
                ("ld_b", ctypes.c_int), ("perm_h", ctypes.c_int), ("block_start", ctypes.c_int),
                ("reserved", ctypes.c_int)]



def job_blocks(j):
    """Grid blocks (64x64 tiles) of one job -- must match prep_batch_kernel (csrc/elementwise.hip)."""
    if j['kind'] == KIND_CONV:
        return ((j['d0'] + 63) // 64) * ((j['d2'] * j['d3'] * j['pad'] + 63) // 64)
    if j['kind'] == KIND_MATRIX:
        return ((j['d0'] + 63) // 64) * ((j['d1'] + 63) // 64)
    if j['kind'] == KIND_STEM:
        return 1
    return (j['d0'] + 4095) // 4096

MAX_JOBS = 1024     # per mr_prep_batch launch


def conv_job(src, strides, dst_krsc, dst_crsk, K, C, R, S, Cpad, ldk):
    sk, sc, sr, ss = strides
    return dict(kind=KIND_CONV, src=src, src2=0, dst_a=dst_krsc, dst_b=dst_crsk, s0=sk, s1=sc, s2=sr, s3=ss, d0=K, d1=C,
                d2=R, d3=S, pad=Cpad, ld_b=ldk, perm_h=0, total=K * R * S * Cpad)


def matrix_job(src, lds, dst_n, ldn, dst_t, ldt, R, C, perm_h):
    return dict(kind=KIND_MATRIX, src=src, src2=0, dst_a=dst_n, dst_b=dst_t, s0=lds, s1=0, s2=0, s3=0, d0=R, d1=C, d2=0,
                d3=0, pad=ldn, ld_b=ldt, perm_h=perm_h, total=R * C)


def bias_job(a, b, dst, R, perm_h):
    return dict(kind=KIND_BIAS, src=a, src2=b, dst_a=dst, dst_b=0, s0=0, s1=0, s2=0, s3=0, d0=R, d1=0, d2=0, d3=0, pad=0,
                ld_b=0, perm_h=perm_h, total=R)


def stem_job(src, strides, dst, cin):
    sk, sc, sr, ss = strides
    return dict(kind=KIND_STEM, src=src, src2=0, dst_a=dst, dst_b=0, s0=sk, s1=sc, s2=sr, s3=ss, d0=64, d1=cin, d2=3,
                d3=3, pad=0, ld_b=0, perm_h=0, total=64 * 32)


def run_job(dt, j):
    """One job through the individual C entry points (the miss path)."""
    if j['kind'] == KIND_CONV:
        call("mr_prep_conv_weight", dt, j['src'], j['s0'], j['s1'], j['s2'], j['s3'], j['dst_a'], j['dst_b'], j['d0'],
             j['d1'], j['d2'], j['d3'], j['pad'], j['ld_b'])
    elif j['kind'] == KIND_STEM:
        call("mr_stem_pack", j['src'], j['s0'], j['s1'], j['s2'], j['s3'], j['dst_a'], j['d1'])
    elif j['kind'] == KIND_MATRIX:
        call("mr_prep_matrix", dt, j['src'], j['s0'], j['dst_a'], j['pad'], j['dst_b'], j['ld_b'], j['d0'], j['d1'],
             j['perm_h'])
    else:
        call("mr_prep_bias", j['src'], j['src2'], j['dst_a'], j['d0'], j['perm_h'])


class _Entry(object):
    __slots__ = ("params", "stamp", "buffers", "jobs", "dtype", "home")

    def __init__(self, params, buffers, jobs, dtype):
        self.params = params
        self.buffers = buffers
        self.jobs = jobs
        self.dtype = dtype
        self.stamp = _stamp(params)
        # the jobs hold RAW source pointers: the entry can only be refreshed while the parameters still live there
        self.home = tuple(p.data_ptr() for p in params if p is not None)

    def at_home(self):
        return self.home == tuple(p.data_ptr() for p in self.params if p is not None)


def _stamp(params):
    return tuple((p.data_ptr(), p._version) for p in params if p is not None)


def prepared(params, key, build, dtype):
    """params: tuple of source Parameters (None entries allowed); key: hashable description of the image;
    build(old_buffers_or_None) -> (buffers, jobs): allocates the buffers when given None (else reuses them) and
    returns the job descriptions (conv_job / matrix_job / bias_job) that fill them.  Returns the buffers."""
    owner = params[0]
    cache = owner.__dict__.get("_mr_prep")
    if cache is None:
        cache = owner.__dict__["_mr_prep"] = {}
    full_key = (key, dtype)
    e = cache.get(full_key)
    if e is not None and e.stamp == _stamp(params) and all(a is b for a, b in zip(e.params, params)):
        return e.buffers
    buffers, jobs = build(e.buffers if e is not None else None)
    dt = dtype_code(dtype)
    for j in jobs:
        run_job(dt, j)
    cache[full_key] = _Entry(tuple(params), buffers, jobs, dtype)
    return buffers


class _Plan(object):
    """Device job table for one set of entries (one per compute dtype)."""

    def __init__(self, entries, device):
        self.signature = tuple(id(e) for e in entries)
        self.entries = entries
        self.tables = []
        by_dtype = {}
        for e in entries:
            by_dtype.setdefault(e.dtype, []).extend(e.jobs)
        chunks = [(dtype, jobs[i:i + MAX_JOBS]) for dtype, jobs in by_dtype.items()
                  for i in range(0, len(jobs), MAX_JOBS)]
        for dtype, jobs in chunks:
            arr = (PrepJob * len(jobs))()
            nblocks = 0
            for slot, j in zip(arr, jobs):
                for name, _ in PrepJob._fields_:
                    if name in j:
                        setattr(slot, name, j[name])
                slot.block_start = nblocks
                nblocks += job_blocks(j)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.tables.append((dtype_code(dtype), host.to(device), len(jobs), nblocks))

    def launch(self, tick=None):
        """tick: the optimizer's device hyper block -- the first launch also advances its step counter (mr_prep_batch)."""
        for dt, table, njobs, nblocks in self.tables:
            call("mr_prep_batch", dt, ptr(table), njobs, nblocks, ptr(tick))
            tick = None
        if tick is not None:
            call("mr_opt_tick", ptr(tick))


def refresh(params, holder, tick=None):
    """Regenerate every recorded image whose sources are in `params` (called by the fused optimizers right after the
    parameter update).  `holder` (the optimizer's per-group dict) keeps the device job table between calls.
    tick: the optimizer's device hyper block; its step counter is advanced by the regeneration launch (or by a one-thread
    launch when there is nothing to regenerate) -- the update kernel itself only reads it."""
    entries = []
    for p in params:
        cache = p.__dict__.get("_mr_prep")
        if cache:
            for k in [k for k, e in cache.items() if not e.at_home()]:
                del cache[k]    # a source parameter was re-pointed (p.data = ...): rebuilt by the next forward
            entries.extend(cache.values())
    if not entries:
        holder.pop('prep_plan', None)
        if tick is not None:
            call("mr_opt_tick", ptr(tick))
        return
    plan = holder.get('prep_plan')
    if plan is None or plan.signature != tuple(id(e) for e in entries):
        plan = holder['prep_plan'] = _Plan(entries, params[0].device)
    plan.launch(tick)
    for e in entries:
        e.stamp = _stamp(e.params)


def invalidate(module_or_params):
    """Drop all prepared images (next forward rebuilds them)."""
    params = module_or_params.parameters() if hasattr(module_or_params, "parameters") else module_or_params
    for p in params:
        p.__dict__.pop("_mr_prep", None)
