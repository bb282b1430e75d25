"""hipGraph capture of a whole training step (forward + loss + backward + fused optimizer update).

The per-step host cost of the eager path (~150 kernel launches through Python autograd, ~5 ms) is comparable to the
GPU time of a CRNN step on MI355X; replaying one captured graph removes it.  Works because every kernel of the path
is launched on torch's current stream through the C ABI (captured like any other launch), the fused optimizers keep
their step counter / hyper-parameters on the device, and the library's only lazily-created state (the zero page) is
created at load time (`mr_init`).

    step = GraphedTrainStep(model_fn, optimizer, static_inputs)   # model_fn(*static_inputs) -> scalar loss tensor
    for batch in loader:
        step.copy_inputs(*batch)      # device-to-device / H2D copies into the static buffers
        loss = step()                 # graph replay; returns the static loss tensor (no host sync)

Data parallel (`grad_sync=`): the step is captured as TWO graphs -- [zero_grad, forward, backward] and
[optimizer update + weight-image refresh] -- with the gradient exchange between them issued eagerly: one in-place
RCCL all-reduce per flat gradient buffer of the fused optimizer (`data_parallel_grad_sync`).  No collective is
captured inside a graph (nothing to go wrong between RCCL and hipGraph), the host cost per step is two graph
launches + one collective, and the exchanged buffer is the optimizer's own contiguous f32 gradient storage (no
bucketing copies).  The price is that the all-reduce is not overlapped with backward: 33 MB at CRNN size is
~0.2-0.3 ms over xGMI against a 4 ms step.  (The eager path -- megreader_amd.apex.parallel -- does overlap, bucket
by bucket, and is what the reference's trainer uses.)
"""
import torch


def no_resident_grid_kernels_beside_collectives(world_size):
    """With more than one rank a collective kernel (RCCL, on its side stream) can hold CUs while the backward pass runs.  The
    one-pass BatchNorm backward (mr_tuning.bn_onepass) passes a barrier that needs its whole grid resident at once: beside a
    collective it would not deadlock (the collective finishes without it) but its resident workgroups would spin until the
    collective frees the CUs.  Data-parallel runs therefore keep the two-launch backward, which is what every multi-rank code
    path was validated with; single-GPU runs keep the one-pass kernel.  Called by the DDP shim and data_parallel_grad_sync."""
    if world_size > 1:
        from . import _lib
        if _lib.get_tuning()["bn_onepass"]:
            _lib.set_tuning(bn_onepass=0)


def data_parallel_grad_sync(optimizer, group=None, average=True, fold=False):
    """Returns a function that averages the gradients of a fused optimizer across the ranks of `group`: one
    in-place all-reduce per flat gradient buffer (apex DDP semantics: sum, then divide by the world size).
    fold=True (opt-in; bench.py and the graphed step use it): the 1 / world factor is applied inside the optimizer's update
    kernel (`optimizer.set_grad_scale`), not by a separate pass over the reduced buffer.  CHANGED GRADIENT SEMANTICS while
    folded: after the sync `p.grad` holds the SUM over the ranks, world_size times the average -- code that reads gradients
    between the sync and `step()` (clip_grad_norm_, gradient logging) must divide by the world size itself, and exactly one
    backward may precede each `step()` (a second one would re-reduce the already summed buffer).  The scale is not part of
    the optimizer's state_dict: re-apply it after loading a checkpoint."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    no_resident_grid_kernels_beside_collectives(world)
    scale_here = average and world > 1
    if scale_here and fold and hasattr(optimizer, "set_grad_scale"):
        optimizer.set_grad_scale(1.0 / world)
        scale_here = False

    def sync():
        for flat in optimizer.flat_grads():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if scale_here:
                flat.mul_(1.0 / world)
    return sync


def broadcast_parameters(module, src=0, group=None):
    """Rank `src`'s parameters and buffers define the model (what apex DDP does at construction)."""
    import torch.distributed as dist
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=group)


def scalar_mean(loss):
    """`loss.mean()` of the reference's train step (trainer.py:124) without the two launches it costs when the loss already is
    a 0-dim tensor (nn.CTCLoss(reduction='mean') and the detector's criterion return scalars): the mean of one element is the
    element, forward and backward."""
    return loss if loss.dim() == 0 else loss.mean()


_LIVE_STEPS = []      # weak references to the GraphedTrainStep objects that are alive


_CAPTURE_STREAMS = {}     # device index -> the stream every GraphedTrainStep of this process warms up and captures on


class GraphedTrainStep(object):
    """One captured training step.

    Several captures of one model / optimizer may be alive at once since round 5.  Rounds 3-4 saw replays of the NEWER graph
    race while an older capture was alive (the attention decoder's loss came back as the sum of its first k steps; with more
    replays an HSA memory-aperture violation).  Root cause: torch caches the parameters' AccumulateGrad nodes for as long as any
    earlier autograd graph lives, each bound to the stream it was created on; every instance used to capture on a stream of its
    own, so the second capture found nodes bound to the FIRST capture's stream and the engine synchronised the two streams
    inside the capture.  All instances now warm up and capture on one stream per device (_CAPTURE_STREAMS, see _capture);
    tools/diag_attn_replay.py, profiles/r05_diag_two_live_captures.txt, tests/test_fpn_attention_gpu.py::
    test_two_live_captured_steps_replay_consistently.  (The process-global scratch -- zero arena, split-reduction workspace --
    is shared by all captures by design: replays on one stream are ordered.)"""

    def __init__(self, loss_fn, optimizer, static_inputs, warmup=3, grad_sync=None):
        import gc
        import weakref
        gc.collect()
        _LIVE_STEPS[:] = [r for r in _LIVE_STEPS if r() is not None]
        _LIVE_STEPS.append(weakref.ref(self))
        self.loss_fn = loss_fn
        self.optimizer = optimizer
        self.inputs = list(static_inputs)
        self.grad_sync = grad_sync     # eager gradient exchange between the two graphs (data parallel), or None
        self.graph = None
        self.graph_update = None
        self.loss = None
        self._one = None
        self._warmup = warmup
        self._capture()

    def _fwd_bwd(self):
        self.optimizer.zero_grad()
        loss = self.loss_fn(*self.inputs)
        # the root gradient (ones) is a constant: created once during warm-up instead of by a fill launch in every step
        one = self._one
        if one is None or one.shape != loss.shape or one.dtype != loss.dtype or one.device != loss.device:
            one = self._one = torch.ones_like(loss, requires_grad=False)
        loss.backward(one)
        return loss

    def _eager(self):
        loss = self._fwd_bwd()
        if self.grad_sync is not None:
            self.grad_sync()
        self.optimizer.step()
        return loss

    def _capture(self):
        # Warm-up (lazy initialisations, allocator pools, flat optimizer buffers) and capture run on ONE dedicated side stream.
        #
        # HAZARD (ROCm 7.0 HIP runtime, found in round 4: tools/diag_capture.py, profiles/r04_diag_capture_crash.txt).  Autograd
        # nodes that outlive a step -- the parameters' AccumulateGrad nodes, which torch caches weakly and therefore re-uses for
        # as long as ANY earlier graph is alive (a caller that still holds the previous step's loss, or a reference cycle not
        # yet collected) -- remember the stream they were created on, and the engine synchronises the producing stream with
        # that stream when a gradient arrives.  Inside a capture such a synchronisation pulls the other stream into the
        # capture; when it is the LEGACY DEFAULT stream, hipStreamEndCapture segfaults (hip::Stream::EndCapture walking its
        # parallel capture streams).  Eager steps on a non-default stream are harmless (a legal fork / join), and so is a
        # default-stream graph that is dead by now.  Therefore: collect garbage first, and callers must not keep the loss (or
        # any other tensor with a grad_fn) of a default-stream step alive across the construction of this object.
        import gc
        import os
        gc.collect()
        # ONE capture stream per device for every GraphedTrainStep of the process (round 5).  The AccumulateGrad nodes of the
        # parameters are cached by torch for as long as any earlier autograd graph lives and stay bound to the stream they
        # were created on; with a fresh stream per instance, a second capture while the first is alive found nodes bound to the
        # FIRST capture's stream, the engine synchronised the two streams inside the capture ("The AccumulateGrad node's
        # stream does not match ..."), and replays of the second graph raced: partial decode-step sums in round 4, an
        # HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION after some tens of replays in round 5 (tools/diag_attn_replay.py,
        # profiles/r05_diag_two_live_captures.txt).  With a shared stream the cached nodes already sit on the capture stream.
        dev = torch.cuda.current_device()
        s = _CAPTURE_STREAMS.get(dev)
        if s is None:
            s = _CAPTURE_STREAMS[dev] = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self._warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        same = os.environ.get("MEGREADER_CAPTURE_STREAM", "same") != "own"    # "own": torch's internal capture stream (A/B)
        kw = {"stream": s} if same else {}
        # With a process group alive, RCCL's watchdog thread polls the events of recent collectives (hipEventQuery).  In the default
        # "global" capture mode a call like that from ANY thread invalidates the capture ("operation failed due to a previous
        # error during capture", seen once in ~10 runs of tests/test_ddp_gpu.py::...grad_sync_single_rank_rccl at the capture of
        # the update graph, right behind the eager all-reduce): only this thread's calls are policed while a group exists.
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                kw["capture_error_mode"] = "thread_local"
        except Exception:
            pass
        # self.loss is the captured step's loss buffer WITHOUT its autograd graph: a retained graph would keep the step's
        # AccumulateGrad nodes alive -- bound to THIS capture's stream -- and a later capture (another GraphedTrainStep on the
        # same parameters) would be pulled onto that stream through them (see the hazard note above; measured: replays of the
        # second graph then intermittently read clobbered buffers, tools/diag_attn_replay.py)
        if self.grad_sync is None:
            with torch.cuda.graph(self.graph, **kw):
                self.loss = self._eager().detach()
            gc.collect()
            return
        with torch.cuda.graph(self.graph, **kw):
            self.loss = self._fwd_bwd().detach()
        gc.collect()
        self.grad_sync()
        torch.cuda.synchronize()       # no collective in flight when the second capture begins
        self.graph_update = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_update, **kw):
            self.optimizer.step()

    def copy_inputs(self, *tensors):
        for dst, src in zip(self.inputs, tensors):
            dst.copy_(src, non_blocking=True)

    def __call__(self):
        # the captured update kernel reads lr / betas / weight decay from device slots: push schedule changes made
        # on the host since the last step (the Python optimizer.step() that normally does it is not re-run)
        self.optimizer.push_hyper()
        self.graph.replay()
        if self.graph_update is not None:
            self.grad_sync()
            self.graph_update.replay()
        return self.loss
