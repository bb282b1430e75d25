// Skinny NT GEMM for the decode loops: C[M,N] = act(A[M,K] * B[N,K]^T + bias), M <= 32 (one or two MFMA row tiles).
//
// The per-step GEMMs of the attention decoder (N = 32 samples: [32 x 512] x [512 x 2048], [32 x 2048] x [2048 x 512], ...)
// are latency chains on the 64 x 64-tile kernel: a handful of workgroups each running K / 64 dependent
// load -> LDS -> MFMA rounds (9.1 us per launch on average, 197 launches per training step:
// profiles/r03_fpn_attention_kernel_stats_v1_decode_loop.csv).  Here a workgroup owns 16 output columns for all rows, its
// four waves split K, every wave fetches its operands as MFMA fragments straight from global memory (16-byte loads, up to
// SK_DEPTH k-chunks = 3 * SK_DEPTH loads in flight, no LDS staging, no barrier in the k-loop), and the four partial tiles
// are summed through LDS.  N / 16 workgroups (128 for the stacked hidden projection) instead of N / 64.
// The weights (<= 2 MB) are L2-resident across the 32 steps; the kernel is bound by one memory round trip per SK_DEPTH chunks.
#include "common.h"
#include "igemm_core.h"
#include "../../include/megreader_hip.h"

namespace mr {

constexpr int SK_DEPTH = 4;

template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_skinny_kernel(const T* __restrict__ A, long long lda,
                                                             const T* __restrict__ B, long long ldb,
                                                             T* __restrict__ C, long long ldc,
                                                             const float* __restrict__ bias, int relu, int M, int N,
                                                             int K) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int CK = 4 * VEC;                       // k per fragment chunk: 32 (bf16) / 16 (f32)
  typedef typename Mma<T>::Frag Frag;
  __shared__ f32x4 red[4][2][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nchunks = (K + CK - 1) / CK;
  const int per_wave = (nchunks + 3) / 4;
  const int c_begin = wave * per_wave, c_end = min(nchunks, c_begin + per_wave);
  const int nrow = n0 + l15;
  const bool n_ok = nrow < N, m0_ok = l15 < M, m1_ok = 16 + l15 < M;
  const T* bp = B + (long long)nrow * ldb + lg * VEC;
  const T* ap0 = A + (long long)l15 * lda + lg * VEC;
  const T* ap1 = A + (long long)(16 + l15) * lda + lg * VEC;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (int c = c_begin; c < c_end; c += SK_DEPTH) {
    uint4 fb[SK_DEPTH], fa0[SK_DEPTH], fa1[SK_DEPTH];
#pragma unroll
    for (int u = 0; u < SK_DEPTH; ++u) {
      const int k = (c + u) * CK;
      const bool k_ok = c + u < c_end && k + lg * VEC < K;     // K is a multiple of VEC: a vector is inside or outside
      fb[u] = (k_ok && n_ok) ? ldg16(bp + k) : z;
      fa0[u] = (k_ok && m0_ok) ? ldg16(ap0 + k) : z;
      fa1[u] = (k_ok && m1_ok) ? ldg16(ap1 + k) : z;
    }
#pragma unroll
    for (int u = 0; u < SK_DEPTH; ++u) {
      Mma<T>::run(acc0, *(const Frag*)&fb[u], *(const Frag*)&fa0[u]);   // D[n][m]: lane = 4 consecutive n of one m
      Mma<T>::run(acc1, *(const Frag*)&fb[u], *(const Frag*)&fa1[u]);
    }
  }
  red[wave][0][lane] = acc0;
  red[wave][1][lane] = acc1;
  __syncthreads();
  if (wave >= 2) return;
  // wave 0 finishes the row tile m = l15, wave 1 the tile m = 16 + l15
  f32x4 v = red[0][wave][lane];
  v += red[1][wave][lane];
  v += red[2][wave][lane];
  v += red[3][wave][lane];
  const int m = wave * 16 + l15;
  const int n = n0 + lg * 4;
  if (m >= M || n >= N) return;
  if (bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (n + q < N) v[q] += bias[n + q];
  }
  if (relu) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
  }
  T* dst = C + (long long)m * ldc + n;
  if (n + 3 < N && (ldc & 3) == 0 && ((((uintptr_t)C) & 15) == 0)) {
    store4(dst, v);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (n + q < N) dst[q] = from_f32<T>(v[q]);
  }
}

#define g_skinny MR_TUNE(gemm_skinny)

// used by mr_gemm_nt (gemm_conv.hip): true when the skinny kernel took the problem
bool gemm_nt_skinny(int dtype, const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc,
                    const float* bias, int relu, int M, int N, int K, hipStream_t stream) {
  if (!g_skinny || M > 32 || M < 1 || K < 1) return false;
  const int grid = (N + 15) / 16;
  if (dtype == MR_F32)
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<float>), dim3(grid), dim3(256), 0, stream, (const float*)A, lda, (const float*)B,
                       ldb, (float*)C, ldc, bias, relu, M, N, K);
  else
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, relu, M, N, K);
  return true;
}

}  // namespace mr

extern "C" {

// A/B (host only): 0 = M <= 32 GEMMs take the general tiled kernels again.  Returns the previous setting.
}  // extern "C"
