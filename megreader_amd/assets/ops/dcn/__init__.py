"""`assets.ops.dcn` on MI355X -- mirror of reference assets/ops/dcn/__init__.py:1-13.

Implemented: ModulatedDeformConv / modulated_deform_conv / ModulatedDeformConvPack (DCNv2, the only variant a
reference model uses: backbones/resnet.py:295-309 `deformable_resnet50`) and DeformConv / DeformConvPack / deform_conv
(v1, on the v2 kernels with a mask of ones).  `deform_conv_cuda` is the extension-level module
(src/deform_conv_cuda.cpp entry points) for the reference's own functions/deform_conv.py.  The deformable PS-RoI pooling
modules (exported by the reference, used by no backbone, decoder or YAML -- SURVEY.md §8 f4) run on csrc/deform_pool.hip;
`deform_pool_cuda` is their extension-level module.
"""
from .deform_conv import (ModulatedDeformConv, ModulatedDeformConvPack, ModulatedDeformConvFunction,  # noqa: F401
                          modulated_deform_conv, DeformConv, DeformConvPack, deform_conv)
from . import deform_conv_cuda  # noqa: F401  (extension-level boundary)


from .deform_pool import (DeformRoIPooling, DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack,  # noqa: F401,E402
                          DeformRoIPoolingFunction, deform_roi_pooling)
from . import deform_pool_cuda  # noqa: F401,E402  (extension-level boundary)

__all__ = ['DeformConv', 'DeformConvPack', 'ModulatedDeformConv', 'ModulatedDeformConvPack', 'DeformRoIPooling',
           'DeformRoIPoolingPack', 'ModulatedDeformRoIPoolingPack', 'deform_conv', 'modulated_deform_conv',
           'deform_roi_pooling']
