"""Make the UNMODIFIED reference tree run on the MI355X kernels.

    import megreader_amd.dropin as dropin
    dropin.install("/path/to/MegReader")      # before `import structure.model` / `import trainer`

or, with zero edits to the reference:   python -m megreader_amd.run train.py experiments/recognition/crnn.yaml ...

What install() does:
  * import shims for non-arithmetic third-party deps that may be missing (megreader_amd.compat);
  * `apex`            -> megreader_amd.apex (RCCL DistributedDataParallel; structure/model.py:6,34, resnet.py:4);
  * `ops`             -> megreader_amd.ops (ctc_loss_2d on HIP; decoders/ctc_decoder2d.py:12);
  * the reference's own `backbones` / `decoders` packages are imported unchanged and the factories this package
    implements are overridden in their namespaces (`backbones.crnn_backbone`, `decoders.CRNNDecoder`, ...), so
    `getattr(backbones, args['backbone'])` (structure/model.py:20-21) resolves to the HIP modules while every
    name that is not on the hot path (detection heads, losses) still comes from the reference.
"""
import importlib
import os
import sys

OVERRIDES = {
    "backbones": ("megreader_amd.backbones", ["crnn_backbone", "resnet18", "resnet34", "resnet50", "resnet101",
                                              "resnet152", "deformable_resnet50", "resnet50dilated_ppm",
                                              "Resnet18FPN", "Resnet34FPN", "Resnet50FPN", "Resnet101FPN",
                                              "Resnet152FPN"]),
    "decoders": ("megreader_amd.decoders", ["CRNNDecoder", "CTCDecoder", "CTCDecoder2D", "AttentionDecoder", "SegDetector"]),
    # evaluation side: the YAMLs name these classes through `package: [structure.representers,
    # structure.measurers, ...]` + `class: CTCRepresenter` (concern/config.py:28-29,57-60), resolved with getattr
    "structure.representers": ("megreader_amd.structure", ["CTCRepresenter", "CTCRepresenter2D",
                                                           "SegDetectorRepresenter"]),
    "structure.measurers": ("megreader_amd.structure", ["SequenceRecognitionMeasurer"]),
}


def fuse_optimizers():
    """`getattr(torch.optim, 'Adam')(parameters, **args)` (reference training/optimizer_scheduler.py:17-22) -> the fused
    flat-buffer optimizers of megreader_amd.optim when every parameter is an fp32 CUDA tensor and the options are ones they
    implement; anything else constructs torch's own optimizer.  `torch.optim.Adam` / `SGD` stay CLASSES -- subclasses of the
    originals whose `__new__` hands eligible constructions to the fused optimizer -- so `class X(torch.optim.SGD)` and
    `isinstance(opt, torch.optim.Adam)` in the reference or in third-party code keep working (a fused instance counts as an
    instance of the alias).  Idempotent.  Returns the names that were aliased."""
    import inspect

    import torch
    from . import optim as _optim

    def _alias(orig, fused, unsupported):
        if getattr(orig, "_mr_original", None) is not None:
            return orig
        allowed = set(inspect.signature(fused.__init__).parameters) - {"self", "params"}

        class _Meta(type(orig)):
            def __instancecheck__(cls, inst):
                return type.__instancecheck__(cls, inst) or (cls.__dict__.get("_mr_original") is orig and
                                                              isinstance(inst, fused))

        class Alias(orig, metaclass=_Meta):
            _mr_original = orig
            _mr_fused = fused

            def __new__(cls, params=None, *args, **kwargs):
                if params is None:         # copy.deepcopy / pickle rebuild instances through cls.__new__(cls)
                    return super().__new__(cls)
                params = list(params)      # may be a generator: consumed once, handed on through the instance
                if cls.__dict__.get("_mr_original") is orig:       # the alias itself, not a user subclass of it
                    flat = [p for g in params for p in g["params"]] if params and isinstance(params[0], dict) else params
                    ok = bool(flat) and all(getattr(p, "is_cuda", False) and p.dtype == torch.float32 for p in flat)
                    ok = ok and not args and not any(kwargs.get(k) for k in unsupported) and set(kwargs) <= allowed
                    if ok:
                        return fused(params, **kwargs)             # not an instance of cls: Python skips __init__
                inst = super().__new__(cls)
                inst.__dict__["_mr_params"] = params
                return inst

            def __init__(self, params, *args, **kwargs):
                super().__init__(self.__dict__.pop("_mr_params", params), *args, **kwargs)

        Alias.__name__ = orig.__name__
        Alias.__qualname__ = orig.__qualname__
        Alias.__module__ = orig.__module__
        Alias.__doc__ = orig.__doc__
        return Alias

    torch.optim.Adam = _alias(torch.optim.Adam, _optim.FusedAdam, ("amsgrad", "foreach", "fused", "capturable",
                                                                   "maximize", "differentiable"))
    torch.optim.SGD = _alias(torch.optim.SGD, _optim.FusedSGD, ("dampening", "nesterov", "foreach", "fused", "maximize",
                                                                "differentiable"))
    return ["Adam", "SGD"]


class _GraphedTrainStep(object):
    """Replacement of `Trainer.train_step` (reference trainer.py:114-143: zero_grad / model.forward(batch, training=True) /
    l.mean() / backward / optimizer.step() / logging every log_interval steps) that replays ONE hipGraph per step.

    The first `eager_steps` calls run the original method (real training steps: they also create every lazily allocated
    buffer); the next call copies the batch into static device tensors and captures the same sequence of calls -- the
    model's own forward, with its `.to(device)` calls now no-ops -- with megreader_amd.runtime.GraphedTrainStep; from then on
    a step is: H2D copies of the batch into the static tensors + one graph replay.  The logging block of the original method
    is re-run on the replayed loss / metrics.  Batches whose shapes differ from the captured ones (last batch of an epoch)
    and optimizers that are not megreader_amd fused ones take the original method."""

    def __init__(self, original, eager_steps=3):
        self.original = original
        self.eager_steps = eager_steps
        self.calls = 0
        self.state = None     # (signature, static batch, graphed step, holder)
        self.disabled = False
        self.mode = None      # "single" | "capture" | "graph2" once captured (_capture)

    @staticmethod
    def _signature(batch):
        import torch
        return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()) if isinstance(v, torch.Tensor))

    def __call__(self, trainer, model, optimizer, batch, epoch, step, **kwargs):
        import torch
        from . import optim as _optim
        from .runtime import GraphedTrainStep, scalar_mean
        self.calls += 1
        usable = (not self.disabled and isinstance(optimizer, _optim._FlatOptimizer) and isinstance(batch, dict)
                  and torch.cuda.is_available() and model.training)
        if not usable or self.calls <= self.eager_steps:
            return self.original(trainer, model, optimizer, batch, epoch=epoch, step=step, **kwargs)
        sig = self._signature(batch)
        if self.state is not None and self.state[0] != sig:
            return self._eager_other_signature(trainer, model, optimizer, batch, epoch, step, kwargs)
        if self.state is None:
            device = getattr(trainer, "device", torch.device("cuda"))
            static = {k: (v.to(device).clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
            holder = {}

            def loss_fn():
                results = model.forward(static, training=True)
                metrics = {}
                if isinstance(results, (tuple, list)) and len(results) == 2:
                    l, _pred = results
                elif isinstance(results, (tuple, list)) and len(results) == 3:
                    l, _pred, metrics = results
                else:
                    l = results
                holder["metrics"] = {k: (v.detach() if hasattr(v, "detach") else v) for k, v in metrics.items()}
                return scalar_mean(l)
            try:
                optimizer.push_hyper()     # no host->device copy may happen inside the capture
                graphed = self._capture(model, optimizer, loss_fn)
            except Exception as e:  # noqa: BLE001 - a model that cannot be captured keeps training eagerly
                torch.cuda.synchronize()
                self.disabled = True
                print("megreader_amd.dropin: hipGraph capture of the training step failed (%s: %s); staying eager" %
                      (type(e).__name__, e), file=sys.stderr)
                return self.original(trainer, model, optimizer, batch, epoch=epoch, step=step, **kwargs)
            self.state = (sig, static, graphed, holder)
        _sig, static, graphed, holder = self.state
        self._upload(batch, static)
        loss = graphed()
        self._log(trainer, loss, holder.get("metrics", {}), epoch, step)
        return loss.detach().clone()     # the graph's loss tensor is overwritten by the next replay

    def _capture(self, model, optimizer, loss_fn):
        """Single process: one graph.  Data parallel (`train.py -d`: structure/model.py:27-36 wraps the model in the apex shim):
        'capture' -- the shim's bucketed all-reduces on its side stream are captured INSIDE the step graph (what bench.py
        measures at N > 1; needs a collective backend that can be captured: RCCL), the 1 / world factor folded into the update
        kernel; or 'graph2' -- [zero_grad, forward, backward] and [update] as two graphs with ONE eager in-place all-reduce of
        the flat gradient buffers between them, the shim's hooks suspended (any backend).  MEGREADER_DDP_GRAPH =
        auto (default: capture on nccl, falling back to graph2) | capture | graph2 | off (eager distributed step)."""
        import torch
        import torch.distributed as dist
        from .apex.parallel import DistributedDataParallel as Shim
        from .runtime import GraphedTrainStep, data_parallel_grad_sync
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        shim = model if isinstance(model, Shim) else None
        if world <= 1 or shim is None:
            if world > 1:
                raise RuntimeError("distributed run without the apex.parallel shim around the model: no gradient exchange to capture")
            self.mode = "single"
            return GraphedTrainStep(loss_fn, optimizer, [], warmup=0)
        want = os.environ.get("MEGREADER_DDP_GRAPH", "auto")
        if want == "off":
            raise RuntimeError("MEGREADER_DDP_GRAPH=off")
        backend = dist.get_backend(shim.group)
        if want == "capture" or (want == "auto" and backend == "nccl"):
            try:
                shim.fold_average_into(optimizer)
                graphed = GraphedTrainStep(loss_fn, optimizer, [], warmup=0)
                self.mode = "capture"
                return graphed
            except Exception as e:  # noqa: BLE001
                shim.unfold_average()
                if want == "capture":
                    raise
                torch.cuda.synchronize()
                print("megreader_amd.dropin: in-graph collective capture failed (%s: %s); two graphs + eager all-reduce" %
                      (type(e).__name__, e), file=sys.stderr)
        shim.suspended = True
        try:
            sync = data_parallel_grad_sync(optimizer, group=shim.group, average=shim.gradient_average, fold=True)
            graphed = GraphedTrainStep(loss_fn, optimizer, [], warmup=0, grad_sync=sync)
        except Exception:
            shim.suspended = False
            optimizer.set_grad_scale(1.0)
            raise
        self.mode = "graph2"
        return graphed

    def _eager_other_signature(self, trainer, model, optimizer, batch, epoch, step, kwargs):
        """A batch whose shapes differ from the captured ones (the partial last batch of a DistributedSampler with drop_last=False,
        data/data_loader.py:40-48) takes the reference's eager step.  In 'graph2' mode the data-parallel shim is SUSPENDED and the
        1 / world factor lives in the optimizer (the captured step exchanges gradients between its two graphs): an eager step
        would then update with LOCAL gradients scaled by 1 / world and the ranks would diverge silently.  For that one
        step the shim is switched back on (its hooks all-reduce during backward) and the fold is undone; both are restored after."""
        shim = getattr(model, "model", None)
        if self.mode != "graph2" or shim is None or not getattr(shim, "suspended", False):
            return self.original(trainer, model, optimizer, batch, epoch=epoch, step=step, **kwargs)
        scale = getattr(optimizer, "_grad_scale", None)
        shim.suspended = False
        optimizer.set_grad_scale(1.0)
        try:
            return self.original(trainer, model, optimizer, batch, epoch=epoch, step=step, **kwargs)
        finally:
            shim.suspended = True
            if scale is not None:
                optimizer.set_grad_scale(scale)

    def _upload(self, batch, static):
        """Host batch -> the graph's static tensors without stalling the host: H2D into one of two staging sets on a copy
        stream (it overlaps the previous step's replay, which is still running), then a device-to-device copy on the
        main stream right in front of the replay.  (The DataLoader of the reference pins its batches, data_loader.py:46;
        pageable batches work too, their H2D copy is just synchronous.)"""
        import torch
        if not hasattr(self, "_copy_stream"):
            self._copy_stream = torch.cuda.Stream()
            self._staging = [{k: torch.empty_like(v) for k, v in static.items() if isinstance(v, torch.Tensor)}
                             for _ in range(2)]
            self._consumed = [None, None]      # event: the main stream has copied staging set i into the static tensors
            self._slot = 0
        i = self._slot
        self._slot ^= 1
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self._copy_stream):
            if self._consumed[i] is not None:
                self._copy_stream.wait_event(self._consumed[i])
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    self._staging[i][k].copy_(v, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        main.wait_event(ready)
        for k, st in self._staging[i].items():
            static[k].copy_(st, non_blocking=True)
        done = torch.cuda.Event()
        done.record(main)
        self._consumed[i] = done

    @staticmethod
    def _log(trainer, loss, metrics, epoch, step):
        exp = getattr(trainer, "experiment", None)
        interval = getattr(getattr(exp, "logger", None), "log_interval", None) if exp is not None else None
        if not interval or step % interval != 0:
            return
        logger = trainer.logger
        if getattr(trainer, "is_main", True):   # trainer.py:133-143 (single-process form: reduce() is the identity there)
            logger.info('step: %6d, epoch: %3d, loss: %.6f, lr: %f' % (step, epoch, loss.mean().item(),
                                                                       getattr(trainer, "current_lr", 0.0)))
            logger.add_scalar('loss', loss, step)
            logger.add_scalar('learning_rate', getattr(trainer, "current_lr", 0.0), step)
            for name, metric in metrics.items():
                logger.add_scalar(name, metric.mean(), step)
                logger.info('%s: %6f' % (name, metric.mean()))
            logger.report_time('Logging')


def accelerate_trainer(trainer_cls, eager_steps=3):
    """Patch `trainer_cls.train_step` (reference trainer.py:114) with the graphed step above.  Returns the wrapper."""
    original = trainer_cls.train_step
    if isinstance(getattr(original, "_mr_graphed", None), _GraphedTrainStep):
        return original._mr_graphed
    wrapper = _GraphedTrainStep(original, eager_steps)

    def train_step(self, model, optimizer, batch, epoch=0, step=0, **kwargs):
        return wrapper(self, model, optimizer, batch, epoch, step, **kwargs)
    train_step._mr_graphed = wrapper
    trainer_cls.train_step = train_step
    return wrapper


def install(reference_root=None, level="plugin", fused_optimizer=False, graph_step=False):
    """fused_optimizer=True: torch.optim.Adam / SGD resolve to the fused flat-buffer optimizers (fuse_optimizers()).
    graph_step=True: the reference's `trainer.Trainer.train_step` replays one captured hipGraph per step
    (accelerate_trainer(); needs fused_optimizer and the reference tree on the path; under `-d` the gradient exchange of the
    apex shim is captured inside the graph, or runs eagerly between two graphs: _GraphedTrainStep._capture).
    Together they give the unchanged `train.py` the step bench.py measures.

    level="plugin" (default): the reference's `ops` package is replaced by megreader_amd.ops (our CTCLoss2DFunction).
    level="extension": the reference's OWN `ops/ctc_2d/ctc_loss_2d.py` is used unchanged and only the pybind11 module it
    binds (`ops.ctc_2d.ctc_2d_csrc`, ops/ctc_2d/ctc_loss_2d.py:3) resolves to the HIP implementation
    (megreader_amd.ops.ctc_2d.ctc_2d_csrc) -- the boundary SURVEY.md §8 b2 names.  Needs the reference tree on the path.
    """
    from . import compat
    compat.install()
    from . import apex as _apex
    sys.modules["apex"] = _apex
    sys.modules["apex.parallel"] = _apex.parallel
    from . import ops as _ops
    if level == "extension":
        from .ops.ctc_2d import ctc_2d_csrc as _csrc
        sys.modules["ops.ctc_2d.ctc_2d_csrc"] = _csrc
        if sys.modules.get("ops") is _ops:
            del sys.modules["ops"]
    else:
        sys.modules.setdefault("ops", _ops)
    # `from assets.ops.dcn import ModulatedDeformConv` (backbones/resnet.py:59-64,129-134): the reference's package
    # imports its CUDA extension at import time, so the HIP mirror is registered under the same dotted name
    from .assets.ops import dcn as _dcn
    import types
    if "assets" not in sys.modules:
        pkg = types.ModuleType("assets")
        pkg.__path__ = []
        sys.modules["assets"] = pkg
    if "assets.ops" not in sys.modules:
        sub = types.ModuleType("assets.ops")
        sub.__path__ = []
        sys.modules["assets.ops"] = sub
        sys.modules["assets"].ops = sub
    sys.modules["assets.ops.dcn"] = _dcn
    sys.modules["assets.ops"].dcn = _dcn
    if reference_root is None:
        reference_root = os.environ.get("MEGREADER_REFERENCE")
    if reference_root:
        reference_root = os.path.abspath(reference_root)
        if reference_root not in sys.path:
            sys.path.insert(0, reference_root)
    # `config.sync_bn` of the reference's OWN config.py (backbones/resnet.py:5,27 `import config` ... `if config.sync_bn:`): an
    # explicit hook, resolved against the reference root -- some unrelated importable module named `config` does not count
    from .backbones import resnet as _resnet
    ref_root_for_config = reference_root

    def _reference_sync_bn():
        if not ref_root_for_config:
            return False
        # the reference's OWN config.py, by path: an unrelated importable module called `config` is neither executed nor read
        cfg = sys.modules.get("config")
        where = os.path.dirname(os.path.abspath(getattr(cfg, "__file__", None) or "")) if cfg is not None else None
        if where != ref_root_for_config:
            path = os.path.join(ref_root_for_config, "config.py")
            if not os.path.isfile(path):
                return False
            import importlib.util
            spec = importlib.util.spec_from_file_location("_megreader_reference_config", path)
            cfg = importlib.util.module_from_spec(spec)
            try:
                spec.loader.exec_module(cfg)
            except Exception:  # noqa: BLE001
                return False
        return bool(getattr(cfg, "sync_bn", False))
    _resnet.set_sync_bn_source(_reference_sync_bn)
    installed = {}
    for pkg, (ours, names) in OVERRIDES.items():
        try:
            ref_mod = importlib.import_module(pkg)
        except ImportError:
            # no reference tree on the path: expose our package under the reference's name
            ref_mod = importlib.import_module(ours)
            sys.modules[pkg] = ref_mod
        our_mod = importlib.import_module(ours)
        for name in names:
            if hasattr(our_mod, name):
                setattr(ref_mod, name, getattr(our_mod, name))
                installed.setdefault(pkg, []).append(name)
    if fused_optimizer:
        installed["torch.optim"] = fuse_optimizers()
    if graph_step:
        try:
            trainer_mod = importlib.import_module("trainer")     # the reference's trainer.py (needs the tree on sys.path)
            accelerate_trainer(trainer_mod.Trainer)
            installed["trainer"] = ["Trainer.train_step"]
        except ImportError as e:
            print("megreader_amd.dropin: graph_step requested but the reference's trainer module is not importable (%s)"
                  % e, file=sys.stderr)
    return installed
