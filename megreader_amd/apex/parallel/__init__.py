"""MI355X replacement for ``apex.parallel`` as used by the reference (structure/model.py:27-36,
backbones/resnet.py:26-30): one process per GPU, gradients averaged with RCCL all-reduce over xGMI.

Design (not a port of apex): parameters are grouped into buckets in reverse registration order (~ the order
backward produces gradients).  A post-accumulate-grad hook marks a parameter ready; when a bucket is complete
its gradients are all-reduced on a side HIP stream while backward keeps running on the main stream.  When the
optimizer keeps gradients in one flat buffer (megreader_amd.optim), a bucket is a contiguous slice of it and is
reduced in place with no flatten/unflatten copies; otherwise the bucket is packed into a staging buffer.  An
end-of-backward callback launches incomplete buckets (parameters that received no gradient, e.g. the unused
``cbr_deepsup`` / ``fc`` weights listed in SURVEY.md §8a), waits for the side stream and scales by 1/world.

xGMI is a point-to-point mesh: large buckets keep RCCL in its bandwidth regime and let it use all seven links, but ONE
bucket means the exchange starts only when the first layer's gradient lands, i.e. after backward.  The bucket size is
therefore min(message_size, total / MIN_BUCKETS): at least four buckets, so the decoder / LSTM / upper-conv gradients
(8 MB pieces for the CRNN's 33 MB) travel while the lower convolutions are still in backward.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


_PAD = 64  # megreader_amd.optim._ALIGN: parameter slices of the flat buffers start on 64-element boundaries
MIN_BUCKETS = 4


def _engine_callback(fn):
    torch.autograd.Variable._execution_engine.queue_callback(fn)


def plan_buckets(sizes, target):
    """Bucket plan over REVERSED registration order (~ the order backward produces gradients): lists of indices into
    `sizes`.  A bucket closes when it reaches `target` elements, or BEFORE a parameter that would push it more than 25 %
    over (one 2.4 M-element conv weight must not drag 1.5 M elements of LSTM gradients with it) -- unless the bucket is still
    under 10 % of the target, which is not worth a collective of its own; for the same reason a tail bucket under 10 % of the
    target is merged into its predecessor."""
    buckets, cur, cur_n = [], [], 0
    for i in reversed(range(len(sizes))):
        if cur and cur_n >= 0.1 * target and cur_n + sizes[i] > 1.25 * target:
            buckets.append(cur)
            cur, cur_n = [], 0
        cur.append(i)
        cur_n += sizes[i]
        if cur_n >= target:
            buckets.append(cur)
            cur, cur_n = [], 0
    if cur:
        if buckets and cur_n < 0.1 * target:
            buckets[-1].extend(cur)
        else:
            buckets.append(cur)
    return buckets


class DistributedDataParallel(nn.Module):
    def __init__(self, module, message_size=8 * 1024 * 1024, delay_allreduce=False, gradient_average=True,
                 process_group=None, min_buckets=MIN_BUCKETS, **_ignored):
        super().__init__()
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before DistributedDataParallel(...)")
        self.module = module
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        from ...runtime import no_resident_grid_kernels_beside_collectives
        no_resident_grid_kernels_beside_collectives(self.world_size)
        self.gradient_average = gradient_average
        self.delay_allreduce = delay_allreduce
        self._params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self._params)
        self.bucket_elems = max(1, min(int(message_size), -(-total // max(1, int(min_buckets)))))
        self._use_side_stream = len(self._params) > 0 and self._params[0].is_cuda
        self._stream = torch.cuda.Stream() if self._use_side_stream else None
        # rank 0's weights and buffers define the model (apex behaviour)
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=process_group)
        self._buckets = [[self._params[i] for i in idx] for idx in
                         plan_buckets([p.numel() for p in self._params], self.bucket_elems)]
        self._bucket_of = {}
        for bi, bucket in enumerate(self._buckets):
            for p in bucket:
                self._bucket_of[p] = bi
        self._launched = [False] * len(self._buckets)
        self._n_launched = self._n_staged = 0
        self.last_backward = None
        # parameters known to receive more than one gradient contribution per backward (weight sharing), filled by
        # `mark_shared(param)`: a parameter is complete after `uses` hook firings, a bucket when ALL its parameters are
        self._uses = {}
        self._fires = {}
        self._pending = []  # (bucket index, flat tensor, staged?, work handle)
        self._callback_queued = False
        self._folded = None          # the fused optimizer that applies 1 / world (fold_average_into), else None
        # True: the hooks do nothing -- the gradient exchange is done elsewhere (the two-graph captured step reduces the flat
        # gradient buffers between its graphs with an eager collective, megreader_amd.dropin / bench.py --ddp-mode graph2)
        self.suspended = False
        self._fold_seen_step = None
        for p in self._params:
            hook = self._make_hook()
            p.register_post_accumulate_grad_hook(hook)
            # parameters whose gradient is accumulated through a gradient sink (megreader_amd.nn.functional.grad_sink)
            # never trigger the autograd hook; the ops call these instead
            if not hasattr(p, "_mr_grad_ready_hooks"):
                p._mr_grad_ready_hooks = []
            p._mr_grad_ready_hooks.append(hook)

    def mark_shared(self, param, uses=2):
        """Declare that `param` is used `uses` times per forward (its gradient is complete only after that many
        accumulations): its bucket is then reduced from the end-of-backward callback instead of the first hook."""
        self._uses[id(param)] = int(uses)

    def unfold_average(self):
        """Undo fold_average_into: the shim scales the reduced buffers itself again."""
        if self._folded is not None:
            if self.gradient_average:
                self._folded.set_grad_scale(1.0)
            self._folded = None
            self._fold_seen_step = None

    def fold_average_into(self, optimizer):
        """OPT-IN: let a fused optimizer (megreader_amd.optim) apply the 1 / world_size of the gradient average inside its
        update kernel: the shim then only sums (no `flat.mul_` pass over the reduced buffers).  Every parameter of this module
        must be owned by `optimizer`.
        Changed gradient semantics while folded: after backward `p.grad` holds the SUM over the ranks (world_size times the
        average apex leaves there) -- gradient clipping / logging between backward and step must account for it; the scale
        lives in the optimizer's device hyper block, not in its state_dict; and exactly ONE backward may precede each
        `optimizer.step()`: a second one would all-reduce the already summed flat buffer again (W * sum(g1) + sum(g2)), so
        the shim raises instead (gradient accumulation needs the default, unfolded mode)."""
        owned = {id(p) for group in optimizer.param_groups for p in group['params']}
        if not all(id(p) in owned for p in self._params):
            raise ValueError("fold_average_into: the optimizer does not own every parameter of the wrapped module")
        if self.gradient_average:
            optimizer.set_grad_scale(1.0 / self.world_size)
            self.gradient_average = False
            self._folded = optimizer
            self._fold_seen_step = None

    def _bucket_complete(self, bi):
        return all(self._fires.get(id(p), 0) >= self._uses.get(id(p), 1) for p in self._buckets[bi])

    # ------------------------------------------------------------------ hooks
    def _make_hook(self):
        def hook(param):
            if self.suspended:       # somebody else exchanges the gradients (two-graph step: runtime.GraphedTrainStep(grad_sync=))
                return
            if not self._callback_queued:
                _engine_callback(self._finalize)
                self._callback_queued = True
            if self.delay_allreduce:
                return
            bi = self._bucket_of[param]
            # readiness is counted per parameter: a shared parameter (mark_shared) is complete only after `uses` firings,
            # and the bucket is launched by whichever parameter completes it LAST (not only by the shared one)
            self._fires[id(param)] = self._fires.get(id(param), 0) + 1
            if not self._launched[bi] and self._bucket_complete(bi):
                self._launch(bi)
        return hook

    def _flat_view(self, bucket):
        """Contiguous in-place view over the bucket's gradients if they sit back to back in memory, else None."""
        grads = [p.grad for p in bucket]
        order = sorted(grads, key=lambda g: g.data_ptr())
        base = order[0]
        lo = base.data_ptr()
        hi = max(g.data_ptr() + g.numel() * g.element_size() for g in order)
        try:
            storage_lo = base.untyped_storage().data_ptr()
            if any(g.untyped_storage().data_ptr() != storage_lo for g in order):
                return None
        except Exception:
            return None
        span = (hi - lo) // base.element_size()
        # the bucket's gradients must TILE the span: consecutive slices separated only by the flat optimizer buffer's
        # alignment padding (< _PAD elements, always zero).  A larger gap means somebody else's gradient lives inside
        # the span (reordered param groups, a frozen parameter, another bucket): reducing the span in place would
        # all-reduce and scale that gradient twice -> staged path instead.
        es = base.element_size()
        end = lo
        for g in order:
            gap = (g.data_ptr() - end) // es
            if gap < 0 or gap >= _PAD:
                return None
            end = g.data_ptr() + g.numel() * es
        off = (lo - storage_lo) // base.element_size()
        return torch.empty(0, dtype=base.dtype, device=base.device).set_(base.untyped_storage(), off, (span,), (1,))

    def _launch(self, bi):
        bucket = [p for p in self._buckets[bi] if p.grad is not None]
        self._launched[bi] = True
        if not bucket:
            return
        flat = self._flat_view(bucket)
        staged = flat is None
        if staged:
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])  # logical (row-major) element order
        if self._use_side_stream:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                flat.record_stream(self._stream)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((bucket, flat, staged, work))
        self._n_launched += 1
        self._n_staged += int(staged)

    def _finalize(self):
        # weight gradients still waiting for a grouped launch (nn.functional._TnDefer): launch them and let their parameters
        # report ready BEFORE the incomplete buckets are closed below -- whichever end-of-backward callback runs first
        from ...nn.functional import flush_deferred_wgrads
        flush_deferred_wgrads()
        if self._folded is not None:
            # (zero_grad() and step() both advance the epoch: a skipped step -- zero_grad, then a new backward -- is legal)
            steps = getattr(self._folded, "_grad_epoch", 0)
            if steps == self._fold_seen_step:
                for _bucket, _flat, _staged, work in self._pending:
                    work.wait()          # collectives the hooks already issued: let them land before giving up the round
                self._reset_round()
                self._n_launched = self._n_staged = 0
                raise RuntimeError("apex.parallel.DistributedDataParallel: a second backward() without optimizer.step() / zero_grad() "
                                   "while the gradient average is folded into the optimizer (fold_average_into): the flat "
                                   "gradient buffer already holds the all-reduced sum and would be reduced again.  Use the "
                                   "default (unfolded) mode for gradient accumulation.")
            self._fold_seen_step = steps
        for bi in range(len(self._buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        scale = 1.0 / self.world_size if self.gradient_average else 1.0
        for bucket, flat, staged, work in self._pending:
            work.wait()  # RCCL: the current (main) stream waits for the collective; host does not block
            self._finish_bucket(bucket, flat, staged, scale)
        self._reset_round()
        # what the last backward did: all-reduces issued, and how many of them went through a staging copy
        self.last_backward = {"all_reduces": self._n_launched, "staged": self._n_staged, "buckets": len(self._buckets)}
        self._n_launched = self._n_staged = 0

    def _reset_round(self):
        self._pending = []
        self._fires = {}
        self._launched = [False] * len(self._buckets)
        self._callback_queued = False

    @staticmethod
    def _finish_bucket(bucket, flat, staged, scale):
        if scale != 1.0:
            flat.mul_(scale)
        if staged:
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view(p.grad.shape))
                off += n

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)


from ...nn.modules import BatchNorm2d as _HipBatchNorm2d  # noqa: E402


class SyncBatchNorm(_HipBatchNorm2d):
    """apex.parallel.SyncBatchNorm as the reference constructs it when config.sync_bn is on (backbones/resnet.py:26-30;
    default False, config.py:14): BatchNorm2d whose training-mode statistics are taken over the batches of every rank of
    `process_group`.  Same parameters / buffers / state_dict keys as nn.BatchNorm2d; eval mode and single-process runs are
    the plain HIP BatchNorm2d.  Kernels and the two small all-reduces: nn/functional.py SyncBatchNormFn."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None,
                 channel_last=False, fuse_relu=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine,
                         track_running_stats=track_running_stats, fuse_relu=fuse_relu)
        self.process_group = process_group
        self.all_reduce = None       # test hook: in-place SUM over the (virtual) ranks of a 1-D f64 tensor

    def forward(self, x, residual=None):
        from ...nn import functional as F
        multi = self.all_reduce is not None or (dist.is_available() and dist.is_initialized() and
                                                dist.get_world_size(self.process_group) > 1)
        if not self.training or not multi:
            return super().forward(x, residual)
        reduce_fn = self.all_reduce
        if reduce_fn is None:
            group = self.process_group

            def reduce_fn(t):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return F.sync_batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps,
                                 self.fuse_relu, residual, self.num_batches_tracked, reduce_fn)
