// Shared device/host helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes everywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MR_MAX_DEVICES 16   /* per-device host-side tables (one node: 8 GPUs) */
#define MR_BN_COPIES 8   /* accumulator copies of the BatchNorm forward statistics (norm_pool.hip bn_reduce_vec_kernel,
                            the NT kernels' statistics epilogue): f64 [MR_BN_COPIES][2][C] */

namespace mr {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { MR_F32 = 0, MR_BF16 = 1 };

enum {
  MR_OK = 0,
  MR_ERR_ARG = 1,      // bad argument (shape / alignment / range)
  MR_ERR_DTYPE = 2,    // unsupported dtype code
  MR_ERR_LAUNCH = 3,   // hip launch error
  MR_ERR_UNSUPPORTED = 4,
};

// last error message, returned by mr_last_error()
void set_error(const char* fmt, ...);

#define MR_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      mr::set_error(__VA_ARGS__);          \
      return mr::MR_ERR_ARG;               \
    }                                      \
  } while (0)

#define MR_CHECK_LAUNCH()                                              \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      mr::set_error("%s:%d hip launch failed: %s", __FILE__, __LINE__, \
                    hipGetErrorString(e__));                           \
      return mr::MR_ERR_LAUNCH;                                        \
    }                                                                  \
  } while (0)

// Phase timer (measurement only; include/megreader_hip.h: mr_phase_timer): entry points that consist of several launches bracket
// their parts with HIP events on the launch stream while it is on, so that bench.py can name the dominant KERNEL of a workload whose
// C-ABI calls are composites (mr_dcn2_bwd2: GEMM + coordinate pass + CSR build + gather + im2col + GEMM).  Off: one relaxed load.
enum { MR_PH_DCN_FWD = 0, MR_PH_DCN_GCOL_GEMM, MR_PH_DCN_COORD, MR_PH_DCN_CSR, MR_PH_DCN_DX, MR_PH_DCN_IM2COL, MR_PH_DCN_WGRAD,
       MR_PH_COUNT };
bool phase_timer_on();
void phase_mark(int id, bool end, double work, hipStream_t stream);   // work: algorithmic bytes (or flops) of the phase
struct PhaseScope {
  int id; hipStream_t s; bool on;
  PhaseScope(int id_, double work, hipStream_t s_) : id(id_), s(s_), on(phase_timer_on()) { if (on) phase_mark(id, false, work, s); }
  ~PhaseScope() { if (on) phase_mark(id, true, 0.0, s); }
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

template <typename T> struct VecOf;             // 16-byte vector of T
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// store 4 consecutive values (address must be 4-element aligned)
__device__ __forceinline__ void store4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  *(bf16x4*)p = o;
}
__device__ __forceinline__ f32x4 load4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
  bf16x4 o = *(const bf16x4*)p;
  f32x4 v;
  v[0] = (float)o[0]; v[1] = (float)o[1]; v[2] = (float)o[2]; v[3] = (float)o[3];
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

// Workgroup barrier that orders LDS traffic only: vector-memory loads / stores in flight STAY in flight (fences restricted to the
// "local" address space compile to s_waitcnt lgkmcnt(0) + s_barrier).  __syncthreads() also waits for vmcnt(0) -- every outstanding
// global access of the wave: on the per-step chain of the persistent kernels that put the HBM latency of prefetches and the write
// acknowledgement of result stores in front of every barrier.  (An inline-asm barrier with a "memory" clobber is NOT the same: the
// compiler drains the vector-memory counter in front of it.)  Only where the waves of a workgroup hand data to each other through
// LDS, never through global memory.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace mr

#include "tuning.h"   // mr_tuning: the library's only process-wide switches (MR_TUNE(field))
