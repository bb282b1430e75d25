"""Import the UNMODIFIED reference (read-only at /root/reference) for golden-vector generation and for the
"does the oracle equal the reference" tests.  Only works in the build container; the GPU box has no reference.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("MEGREADER_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "backbones"))


def import_reference(ops_module=None, dcn_module=None):
    """ops_module: object to expose as the top-level `ops` package (the reference's own `ops` imports its CUDA-only
    extension and cannot be imported on CPU; decoders/ctc_decoder2d.py:12 does `from ops import ctc_loss_2d`).

    Put the reference on sys.path (after installing import shims for its non-arithmetic deps and an inert
    `apex` namespace -- the CPU path never touches apex: structure/model.py:27-36 only uses it under -d)."""
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from megreader_amd import compat
    compat.install()
    if "apex" not in sys.modules:
        apex = types.ModuleType("apex")
        apex.parallel = types.ModuleType("apex.parallel")
        sys.modules["apex"] = apex
        sys.modules["apex.parallel"] = apex.parallel
    if ops_module is not None:
        sys.modules["ops"] = ops_module
    if dcn_module is not None:  # `from assets.ops.dcn import ModulatedDeformConv` (backbones/resnet.py:129-134)
        for name in ("assets", "assets.ops"):
            if name not in sys.modules:
                pkg = types.ModuleType(name)
                pkg.__path__ = []
                sys.modules[name] = pkg
        sys.modules["assets.ops.dcn"] = dcn_module
    for name in ("backbones", "decoders", "structure", "concern", "config"):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__file__", "").startswith(REF_ROOT):
            raise RuntimeError("module %r is already imported from %s" % (name, getattr(mod, "__file__", "?")))
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import structure.model  # noqa: F401
    import backbones  # noqa: F401
    import decoders  # noqa: F401
    return sys.modules["structure.model"]
