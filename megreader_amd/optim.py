"""Fused optimizers over one flat fp32 buffer (one kernel launch per step instead of ~5 per parameter tensor).

``FusedAdam`` / ``FusedSGD`` follow torch.optim.Adam / torch.optim.SGD semantics (reference: optimizers are created
by name at training/optimizer_scheduler.py:17-22; recognition YAMLs use Adam lr 1e-3, the DB detector SGD
momentum 0.9 wd 1e-4).  Parameters and gradients are re-homed into contiguous flat buffers (views keep their
shapes/strides), so ``zero_grad`` is one memset, the update is one kernel, and the data-parallel gradient
all-reduce (megreader_amd.apex.parallel) can work on contiguous bucket slices without copies.

The step counter and hyper-parameters live in a small device tensor, so a captured hipGraph replay of the
training step advances bias correction correctly.
"""
import torch

from ._lib import call, ptr
from .nn import prep
from .nn.functional import ZeroArena, zero_segments

_ALIGN = 64  # elements; keeps every parameter slice 256-byte aligned


class _FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._flat = None  # per group: dict(p=, g=, s1=, s2=, hyper=, n=)
        self._py_steps = 0  # host-side count of step() calls
        # advanced by zero_grad() AND step(): "the flat gradient buffer was cleared / consumed since then" (the DDP shim's
        # one-backward-per-round check while the gradient average is folded into the update kernel)
        self._grad_epoch = 0

    # ------------------------------------------------------------------ flat storage
    def _materialize(self):
        self._flat = []
        for group in self.param_groups:
            params = [p for p in group['params'] if p.requires_grad]
            if not params:
                self._flat.append(None)
                continue
            dev = params[0].device
            if dev.type != 'cuda':
                raise NotImplementedError("megreader_amd fused optimizers run on the GPU only (no CPU fallback)")
            offs, total = [], 0
            for p in params:
                if p.dtype != torch.float32:
                    raise TypeError("fused optimizers expect fp32 master parameters")
                offs.append(total)
                total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
            flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
            for p, off in zip(params, offs):
                n = p.numel()
                dense = p.data.is_contiguous() or p.data.is_contiguous(memory_format=torch.channels_last)
                src = p.data if dense else p.data.contiguous()
                view = flat_p[off:off + n].as_strided(src.shape, src.stride())
                view.copy_(src)
                p.data = view
                gview = flat_g[off:off + n].as_strided(src.shape, src.stride())
                if p.grad is not None:
                    gview.copy_(p.grad)
                p.grad = gview
                # gradient sink (megreader_amd.nn.functional.grad_sink): accumulating backward kernels write here
                p._mr_grad_sink = gview
            self._flat.append({'p': flat_p, 'g': flat_g, 'params': params, 'offs': offs, 'n': total,
                               's1': torch.zeros(total, dtype=torch.float32, device=dev),
                               's2': torch.zeros(total, dtype=torch.float32, device=dev),
                               'hyper': torch.zeros(8, dtype=torch.float32, device=dev), 'hyper_host': None})
            # operand images prepared by an earlier forward (before the parameters moved into the flat buffer) hold
            # jobs that read the OLD storage: drop them, the next forward rebuilds them from the new home
            prep.invalidate(params)

    def flat_grads(self):
        """List of flat gradient buffers (one per param group); used by the DDP shim for zero-copy buckets."""
        if self._flat is None:
            self._materialize()
        return [f['g'] for f in self._flat if f is not None]

    def zero_grad(self, set_to_none=False):
        """One memset per group.  Gradients stay attached (views of the flat buffer) unless set_to_none=True, which
        detaches `.grad` (torch semantics: the next backward goes through autograd's accumulation, and `step()`
        folds the result back into the flat buffer).  The flat buffers, the optimizer state (moments, momentum, the
        device step counter) and the parameters' home are NEVER dropped once created."""
        if self._flat is None:
            self._materialize()
        self._grad_epoch += 1
        segments, devices = [], []
        for f in self._flat:
            if f is not None:
                segments.append((f['g'].data_ptr(), f['g'].numel() * 4))      # (padded to 64 elements: whole 16-byte vectors)
                if f['g'].device not in devices:
                    devices.append(f['g'].device)
                if set_to_none:
                    for p in f['params']:
                        p.grad = None     # grad_sink() is inactive while .grad is None; _sync_views re-attaches
        for dev in devices:     # pre-zeroed scratch arena (BatchNorm reductions, LSTM exchange rings): cleared by the same launch
            used = ZeroArena.rewind(dev)
            if used is not None:
                segments.append(used)
        if segments:
            with torch.cuda.device(devices[0]):
                zero_segments(segments)

    def _sync_views(self, f):
        for p, off in zip(f['params'], f['offs']):
            n = p.numel()
            lo = f['g'].data_ptr() + off * 4
            if p.grad is None:
                # grad_sink() is inactive while .grad is None, so nothing was accumulated for this parameter
                f['g'][off:off + n].zero_()
                p.grad = f['g'][off:off + n].as_strided(p.shape, p.stride())
                p._mr_grad_sink = p.grad
            elif p.grad.data_ptr() != lo:
                view = f['g'][off:off + n].as_strided(p.shape, p.stride())
                view.copy_(p.grad)
                p.grad = view
                p._mr_grad_sink = view

    def _hyper_values(self, group):
        raise NotImplementedError

    def _launch(self, f):
        raise NotImplementedError

    def _push_hyper(self, group, f):
        vals = self._hyper_values(group)
        if vals != f['hyper_host']:
            f['hyper'][:5].copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=False)
            f['hyper_host'] = vals

    def set_grad_scale(self, scale):
        """Scale applied to the raw gradients inside the update kernel (device slot hyper[6]).  Data parallel: the DDP shim
        (megreader_amd.apex.parallel.DistributedDataParallel.fold_average_into) sets 1 / world_size here and stops
        scaling the all-reduced flat buffer itself -- one pass over the gradients less per step."""
        if self._flat is None:
            self._materialize()
        self._grad_scale = float(scale)
        for f in self._flat:
            if f is not None:
                f['hyper'][6:7].fill_(float(scale))

    def push_hyper(self):
        """Copy changed hyper-parameters (lr schedule: trainer.py:43-47,81 `update_learning_rate`) to their device
        slots.  `step()` does this itself; a captured hipGraph replay (megreader_amd.runtime.GraphedTrainStep) never
        re-runs the Python `step()`, so it calls this before every replay."""
        if self._flat is None:
            return
        for group, f in zip(self.param_groups, self._flat):
            if f is not None:
                self._push_hyper(group, f)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._flat is None:
            self._materialize()
        self._py_steps += 1
        self._grad_epoch += 1
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            self._sync_views(f)
            self._push_hyper(group, f)
            self._launch(f)
            # the update went through raw pointers (no autograd version bump): regenerate the compute-dtype
            # operand images of these parameters in one launch
            prep.refresh(f['params'], f, tick=f['hyper'])     # ... and advance the device step counter
        return loss


class FusedAdam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference configs")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _hyper_values(self, group):
        return [float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']),
                float(group['weight_decay'])]

    def _launch(self, f):
        call("mr_adam_step", ptr(f['p']), ptr(f['g']), ptr(f['s1']), ptr(f['s2']), f['n'], ptr(f['hyper']))


class FusedSGD(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, dampening=0, nesterov=False):
        if dampening or nesterov:
            raise NotImplementedError("dampening / nesterov are not used by the reference configs")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    def _hyper_values(self, group):
        return [float(group['lr']), float(group['momentum']), 0.0, 0.0, float(group['weight_decay'])]

    def _launch(self, f):
        call("mr_sgd_step", ptr(f['p']), ptr(f['g']), ptr(f['s1']), f['n'], ptr(f['hyper']))
