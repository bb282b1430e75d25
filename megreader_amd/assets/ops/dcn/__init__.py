"""`assets.ops.dcn` on MI355X -- mirror of reference assets/ops/dcn/__init__.py:1-13.

Implemented: ModulatedDeformConv / modulated_deform_conv / ModulatedDeformConvPack (DCNv2, the only variant a
reference model uses: backbones/resnet.py:295-309 `deformable_resnet50`) and DeformConv / DeformConvPack / deform_conv
(v1, on the v2 kernels with a mask of ones).  `deform_conv_cuda` is the extension-level module
(src/deform_conv_cuda.cpp entry points) for the reference's own functions/deform_conv.py.  The deformable PS-RoI pooling
modules are exported by the reference but used by no backbone, decoder or YAML: they raise
NotImplementedError here.
"""
from .deform_conv import (ModulatedDeformConv, ModulatedDeformConvPack, ModulatedDeformConvFunction,  # noqa: F401
                          modulated_deform_conv, DeformConv, DeformConvPack, deform_conv)
from . import deform_conv_cuda  # noqa: F401  (extension-level boundary)


def _unused(name):
    class _Unused(object):
        def __init__(self, *a, **k):
            raise NotImplementedError("%s is exported by the reference's assets.ops.dcn but used by no model; it is "
                                      "out of scope of the MI355X hot path (SURVEY.md §2b)" % name)
    _Unused.__name__ = name
    return _Unused


DeformRoIPooling = _unused("DeformRoIPooling")
DeformRoIPoolingPack = _unused("DeformRoIPoolingPack")
ModulatedDeformRoIPoolingPack = _unused("ModulatedDeformRoIPoolingPack")


def deform_roi_pooling(*a, **k):
    raise NotImplementedError("deform_roi_pooling is not on the hot path (SURVEY.md §2b)")


__all__ = ['DeformConv', 'DeformConvPack', 'ModulatedDeformConv', 'ModulatedDeformConvPack', 'DeformRoIPooling',
           'DeformRoIPoolingPack', 'ModulatedDeformRoIPoolingPack', 'deform_conv', 'modulated_deform_conv',
           'deform_roi_pooling']
