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
    "decoders": ("megreader_amd.decoders", ["CRNNDecoder", "CTCDecoder2D", "AttentionDecoder", "SegDetector"]),
    # evaluation side: the YAMLs name these classes through `package: [structure.representers,
    # structure.measurers, ...]` + `class: CTCRepresenter` (concern/config.py:28-29,57-60), resolved with getattr
    "structure.representers": ("megreader_amd.structure", ["CTCRepresenter", "CTCRepresenter2D",
                                                           "SegDetectorRepresenter"]),
    "structure.measurers": ("megreader_amd.structure", ["SequenceRecognitionMeasurer"]),
}


def install(reference_root=None, level="plugin"):
    """level="plugin" (default): the reference's `ops` package is replaced by megreader_amd.ops (our CTCLoss2DFunction).
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
    return installed
