// Test infrastructure (oracle/): force-included in front of the reference's OWN extension sources
// (/root/reference/assets/ops/dcn/src/*.cpp, *.cu and /root/reference/ops/ctc_2d/csrc/** -- compiled where they lie, never copied) so that they build with hipcc
// against this image's PyTorch 2.10 / ROCm.  Nothing here implements any part of the operators: it only maps names the
// 2019-era sources use onto their current spellings.
//   AT_CHECK                          -> TORCH_CHECK                      (c10/util/Exception.h dropped the old name)
//   AT_DISPATCH_*(tensor.type(), ...) -> needs ::detail::scalar_type(const DeprecatedTypeProperties&), removed from ATen/Dispatch.h
//   cudaError_t / cudaGetLastError / cudaSuccess / cudaGetErrorString -> the HIP runtime's names
//   max / min (float, double)        -> the mixed overloads CUDA's math headers have and HIP's lack
//   <THC/THCAtomics.cuh>              -> oracle/ref_compat/THC/THCAtomics.cuh (ATen/hip/Atomic.cuh: atomicAdd for Half / double)
#pragma once
#include <hip/hip_runtime.h>
#include <torch/extension.h>
#include <ATen/ATen.h>
#include <ATen/core/DeprecatedTypeProperties.h>

#ifndef AT_CHECK
#define AT_CHECK TORCH_CHECK
#endif

namespace detail {
inline at::ScalarType scalar_type(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
}  // namespace detail

// CUDA's math headers overload max / min for mixed float / double arguments (math_functions.hpp: the float is widened); HIP's do
// not, and deform_pool_cuda_kernel.cu:89-90,129-130 calls max(float, 0.1).  Same semantics as CUDA's.
__host__ __device__ inline double max(float a, double b) { return fmax((double)a, b); }
__host__ __device__ inline double max(double a, float b) { return fmax(a, (double)b); }
__host__ __device__ inline double min(float a, double b) { return fmin((double)a, b); }
__host__ __device__ inline double min(double a, float b) { return fmin(a, (double)b); }

#define cudaError_t hipError_t
#define cudaGetLastError hipGetLastError
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
