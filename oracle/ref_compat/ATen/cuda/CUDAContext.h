// the ROCm build of PyTorch ships this header un-hipified (it includes <cuda_runtime_api.h>); its hipified twin is
// ATen/hip/HIPContext.h.  ops/ctc_2d/csrc/cuda/ctc2d_cuda.cu:3 includes it (see ../../compat.h).
#pragma once
#include <ATen/hip/HIPContext.h>
