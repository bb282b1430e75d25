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

// SKD = k-chunks in flight per wave (3 * SKD 16-byte loads): 4 covers K <= 512 (bf16) in one memory round trip; K = 1536 / 2048
// (the backward GEMMs of the decode loop) took 3 / 4 dependent trips with it and take 2 with 8 (round 4, mr_tuning.skinny_depth).
template <typename T, int SKD = SK_DEPTH>
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
  for (int c = c_begin; c < c_end; c += SKD) {
    uint4 fb[SKD], fa0[SKD], fa1[SKD];
#pragma unroll
    for (int u = 0; u < SKD; ++u) {
      const int k = (c + u) * CK;
      const bool k_ok = c + u < c_end && k + lg * VEC < K;     // K is a multiple of VEC: a vector is inside or outside
      fb[u] = (k_ok && n_ok) ? ldg16(bp + k) : z;
      fa0[u] = (k_ok && m0_ok) ? ldg16(ap0 + k) : z;
      fa1[u] = (k_ok && m1_ok) ? ldg16(ap1 + k) : z;
    }
#pragma unroll
    for (int u = 0; u < SKD; ++u) {
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

// ---------------------------------------------------------------------------------------------------------------------
// Decode-step fusions (round 4): the element-wise GRU kernels of the attention decoder ride in the epilogue of the skinny
// GEMM next to them, and the output layer + log-softmax + NLL + arg-max feedback is one kernel.  Per decode step the forward
// chain was  GEMM(h) -> attention -> GEMM(context) -> GRU gates -> GEMM(out) -> NLL  (6 launches), now
// GEMM(h) -> attention -> [GEMM(context) + GRU gates] -> [out + NLL]  (4); backward  GRU' -> GEMM -> attention' -> GEMM  (4)
// becomes  [GEMM + GRU'] -> GEMM -> attention'  (3).  Reference: decoders/attention_decoder.py:92-115,195-231 (one
// AttentionRNNCell step: nn.GRUCell, nn.Linear out, log_softmax, NLLLoss, topk(1) feedback).
// ---------------------------------------------------------------------------------------------------------------------

// gi_c = A[M,K] * B[3H,K]^T for the 16 hidden units of this workgroup (three 16-column MFMA tiles: the r, z, n rows of the
// same units), then the GRU cell of those units for every sample (the math of gru_fwd2_kernel, with the context part of the
// input gates still in f32):
//   r = s(gi_w[r] + gi_c[r] + gh[r]),  z = s(gi_w[z] + gi_c[z] + gh[z]),  n = tanh(gi_w[n] + gi_c[n] + r * gh[n]),
//   h' = (1 - z) n + z h;     gi_w = row idx[m] of the word table G,  gh = the hidden projection (bias included)
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_skinny_gru_fwd_kernel(
    const T* __restrict__ A, long long lda, const T* __restrict__ B, long long ldb, const T* __restrict__ G, long long ldG,
    const long long* __restrict__ idx, const T* __restrict__ gh, long long ldgh, const T* __restrict__ h,
    T* __restrict__ hnew, float* __restrict__ save, int M, int H, int K) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int CK = 4 * VEC;
  typedef typename Mma<T>::Frag Frag;
  __shared__ f32x4 red[4][3][2][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int u0 = blockIdx.x * 16;
  const int nchunks = (K + CK - 1) / CK;
  const int per_wave = (nchunks + 3) / 4;
  const int c_begin = wave * per_wave, c_end = min(nchunks, c_begin + per_wave);
  const bool u_ok = u0 + l15 < H, m0_ok = l15 < M, m1_ok = 16 + l15 < M;
  const T* bp = B + (long long)(u0 + l15) * ldb + lg * VEC;      // + g * H * ldb for gate g
  const long long gstride = (long long)H * ldb;
  const T* ap0 = A + (long long)l15 * lda + lg * VEC;
  const T* ap1 = A + (long long)(16 + l15) * lda + lg * VEC;
  f32x4 acc[3][2];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g][0] = acc[g][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  constexpr int DEPTH = 6;      // K = 552: 5 chunks per wave, one memory round trip (5 x 5 16-byte loads in flight)
  for (int c = c_begin; c < c_end; c += DEPTH) {
    uint4 fb[DEPTH][3], fa0[DEPTH], fa1[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int k = (c + u) * CK;
      const bool k_ok = c + u < c_end && k + lg * VEC < K;
#pragma unroll
      for (int g = 0; g < 3; ++g) fb[u][g] = (k_ok && u_ok) ? ldg16(bp + g * gstride + k) : z4;
      fa0[u] = (k_ok && m0_ok) ? ldg16(ap0 + k) : z4;
      fa1[u] = (k_ok && m1_ok) ? ldg16(ap1 + k) : z4;
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        Mma<T>::run(acc[g][0], *(const Frag*)&fb[u][g], *(const Frag*)&fa0[u]);   // D[unit][m]: lane = 4 consecutive units of one m
        Mma<T>::run(acc[g][1], *(const Frag*)&fb[u][g], *(const Frag*)&fa1[u]);
      }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    red[wave][g][0][lane] = acc[g][0];
    red[wave][g][1][lane] = acc[g][1];
  }
  __syncthreads();
  if (wave >= 2) return;
  const int m = wave * 16 + l15;
  const int j0 = u0 + lg * 4;
  if (m >= M) return;
  f32x4 v[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    v[g] = red[0][g][wave][lane];
    v[g] += red[1][g][wave][lane];
    v[g] += red[2][g][wave][lane];
    v[g] += red[3][g][wave][lane];
  }
  const T* ga = G + (idx ? idx[m] : (long long)m) * ldG;
  const T* hh = gh + (long long)m * ldgh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = j0 + q;
    if (j >= H) break;
    const float ir = to_f32(ga[j]) + v[0][q], iz = to_f32(ga[H + j]) + v[1][q], in_ = to_f32(ga[2 * H + j]) + v[2][q];
    const float hr = to_f32(hh[j]), hz = to_f32(hh[H + j]), hn = to_f32(hh[2 * H + j]);
    const float r = sigmoidf_(ir + hr), z = sigmoidf_(iz + hz);
    const float nn_ = tanhf_(in_ + r * hn);
    const float hp = to_f32(h[(long long)m * H + j]);
    hnew[(long long)m * H + j] = from_f32<T>((1.f - z) * nn_ + z * hp);
    const long long b3 = (long long)m * 3 * H;
    save[b3 + j] = r;
    save[b3 + H + j] = z;
    save[b3 + 2 * H + j] = nn_;
  }
}

// dh_a = A[M,K] * B[H,K]^T (the gradient the NEXT step's stacked hidden projection sends to h') for the 16 hidden units of this
// workgroup, then the GRU backward of those units (the math of gru_bwd2_kernel with dh_a still in f32):
//   g = dh_a + dh_b + dh_c;  dgi = (dpre_r, dpre_z, dpre_n),  dgh = (dpre_r, dpre_z, dpre_n * r),  dh_prev = g * z
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_skinny_gru_bwd_kernel(
    const T* __restrict__ A, long long lda, const T* __restrict__ B, long long ldb, const T* __restrict__ dh_b,
    const T* __restrict__ dh_c, const float* __restrict__ save, const T* __restrict__ gh, long long ldgh,
    const T* __restrict__ h, T* __restrict__ dgi, T* __restrict__ dgh, long long lddgh, T* __restrict__ dh_prev, int M, int H,
    int K) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int CK = 4 * VEC;
  typedef typename Mma<T>::Frag Frag;
  __shared__ f32x4 red[4][2][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const int u0 = blockIdx.x * 16;
  const int nchunks = (K + CK - 1) / CK;
  const int per_wave = (nchunks + 3) / 4;
  const int c_begin = wave * per_wave, c_end = min(nchunks, c_begin + per_wave);
  const bool u_ok = u0 + l15 < H, m0_ok = l15 < M, m1_ok = 16 + l15 < M;
  const T* bp = B + (long long)(u0 + l15) * ldb + lg * VEC;
  const T* ap0 = A + (long long)l15 * lda + lg * VEC;
  const T* ap1 = A + (long long)(16 + l15) * lda + lg * VEC;
  constexpr int BD = 8;         // K = 2048: 16 chunks per wave, two round trips
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  for (int c = c_begin; c < c_end; c += BD) {
    uint4 fb[BD], fa0[BD], fa1[BD];
#pragma unroll
    for (int u = 0; u < BD; ++u) {
      const int k = (c + u) * CK;
      const bool k_ok = c + u < c_end && k + lg * VEC < K;
      fb[u] = (k_ok && u_ok) ? ldg16(bp + k) : z4;
      fa0[u] = (k_ok && m0_ok) ? ldg16(ap0 + k) : z4;
      fa1[u] = (k_ok && m1_ok) ? ldg16(ap1 + k) : z4;
    }
#pragma unroll
    for (int u = 0; u < BD; ++u) {
      Mma<T>::run(acc0, *(const Frag*)&fb[u], *(const Frag*)&fa0[u]);
      Mma<T>::run(acc1, *(const Frag*)&fb[u], *(const Frag*)&fa1[u]);
    }
  }
  red[wave][0][lane] = acc0;
  red[wave][1][lane] = acc1;
  __syncthreads();
  if (wave >= 2) return;
  const int m = wave * 16 + l15;
  const int j0 = u0 + lg * 4;
  if (m >= M) return;
  f32x4 v = red[0][wave][lane];
  v += red[1][wave][lane];
  v += red[2][wave][lane];
  v += red[3][wave][lane];
  const long long b3 = (long long)m * 3 * H;
  T* dg = dgh + (long long)m * lddgh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = j0 + q;
    if (j >= H) break;
    const long long i = (long long)m * H + j;
    const float r = save[b3 + j], z = save[b3 + H + j], nn_ = save[b3 + 2 * H + j];
    const float hn = to_f32(gh[(long long)m * ldgh + 2 * H + j]);
    const float hp = to_f32(h[i]);
    float g = v[q];
    if (dh_b) g += to_f32(dh_b[i]);
    if (dh_c) g += to_f32(dh_c[i]);
    const float dn = g * (1.f - z);
    const float dz = g * (hp - nn_);
    const float dpre_n = dn * (1.f - nn_ * nn_);
    const float dr = dpre_n * hn;
    const float dpre_r = dr * r * (1.f - r);
    const float dpre_z = dz * z * (1.f - z);
    dgi[b3 + j] = from_f32<T>(dpre_r);
    dgi[b3 + H + j] = from_f32<T>(dpre_z);
    dgi[b3 + 2 * H + j] = from_f32<T>(dpre_n);
    dg[j] = from_f32<T>(dpre_r);
    dg[H + j] = from_f32<T>(dpre_z);
    dg[2 * H + j] = from_f32<T>(dpre_n * r);
    dh_prev[i] = from_f32<T>(g * z);
  }
}

// Output layer + log-softmax + masked NLL + arg-max + the word fed to the next step, one workgroup per sample: logits[c] =
// h[n, :] . W[c, :] + b[c] with the classes dealt over the four waves and K over the lanes (wave reduction), kept in f32 in
// LDS; wave 0 then does what nll_step_fwd_kernel does on the stored logits.  C <= 256.
template <typename T>
__global__ __launch_bounds__(256) void out_nll_fwd_kernel(const T* __restrict__ h, long long ldh, const T* __restrict__ W,
                                                          long long ldw, const float* __restrict__ bias,
                                                          const long long* __restrict__ target, long long tstride,
                                                          const float* __restrict__ mask, float* __restrict__ lp,
                                                          float* __restrict__ loss, long long* __restrict__ argmax,
                                                          const int* __restrict__ feed_flag, long long* __restrict__ feed_idx,
                                                          int C, int K, int accumulate) {
  constexpr int VEC = VecOf<T>::N;
  __shared__ float logit[256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = blockIdx.x;
  const T* hr = h + (long long)n * ldh;
  // a wave owns the classes wave, wave + 4, ...: MAXI of them per pass, their weight vectors all in flight before the first
  // FMA (one class at a time was a chain of C / 4 dependent L2 round trips: slower than the GEMM + NLL launches it replaced)
  constexpr int MAXI = 10;
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  for (int c0 = wave; c0 < C; c0 += 4 * MAXI) {
    float a[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) a[i] = 0.f;
    for (int k = lane * VEC; k < K; k += 64 * VEC) {           // K is a multiple of VEC (the padded weight image)
      const uint4 hv = ldg16(hr + k);
      uint4 wv[MAXI];
#pragma unroll
      for (int i = 0; i < MAXI; ++i) wv[i] = (c0 + 4 * i < C) ? ldg16(W + (long long)(c0 + 4 * i) * ldw + k) : z4;
      const T* ph = (const T*)&hv;
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        const T* pw = (const T*)&wv[i];
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[i] += to_f32(ph[j]) * to_f32(pw[j]);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const float t = wave_sum(a[i]);
      const int c = c0 + 4 * i;
      if (lane == 0 && c < C) logit[c] = t + (bias ? bias[c] : 0.f);
    }
  }
  __syncthreads();
  if (wave != 0) return;
  float mx = -INFINITY;
  int am = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float x = logit[c];
    if (x > mx) { mx = x; am = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {     // wave arg-max with first-index tie break
    const float omx = __shfl_xor(mx, o, 64);
    const int oam = __shfl_xor(am, o, 64);
    if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
  }
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(logit[c] - mx);
  se = wave_sum(se);
  const float lz = mx + logf(se);
  for (int c = lane; c < C; c += 64) lp[(long long)n * C + c] = logit[c] - lz;
  if (lane == 0) {
    if (argmax) argmax[n] = am;
    const long long tg = target[(long long)n * tstride];
    if (feed_idx) feed_idx[n] = (feed_flag && *feed_flag != 0) ? tg : (long long)am;
    if (loss) {
      const float l = -(logit[tg] - lz) * (mask ? mask[n] : 1.f);
      loss[n] = accumulate ? loss[n] + l : l;
    }
  }
}

#define g_skinny MR_TUNE(gemm_skinny)

// used by mr_gemm_nt (gemm_conv.hip): true when the skinny kernel took the problem
bool gemm_nt_skinny(int dtype, const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc,
                    const float* bias, int relu, int M, int N, int K, hipStream_t stream) {
  if (!g_skinny || M > 32 || M < 1 || K < 1) return false;
  const int grid = (N + 15) / 16;
  const int ck = dtype == MR_F32 ? 16 : 32;
  const int per_wave = ((K + ck - 1) / ck + 3) / 4;
  const int depth = MR_TUNE(skinny_depth);
  const bool deep = depth == 8 || (depth == 0 && per_wave > 4);
  if (dtype == MR_F32) {
    if (deep)
      hipLaunchKernelGGL((gemm_nt_skinny_kernel<float, 8>), dim3(grid), dim3(256), 0, stream, (const float*)A, lda,
                         (const float*)B, ldb, (float*)C, ldc, bias, relu, M, N, K);
    else
      hipLaunchKernelGGL((gemm_nt_skinny_kernel<float>), dim3(grid), dim3(256), 0, stream, (const float*)A, lda,
                         (const float*)B, ldb, (float*)C, ldc, bias, relu, M, N, K);
  } else {
    if (deep)
      hipLaunchKernelGGL((gemm_nt_skinny_kernel<bf16_t, 8>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)A, lda,
                         (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, relu, M, N, K);
    else
      hipLaunchKernelGGL((gemm_nt_skinny_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)A, lda,
                         (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, relu, M, N, K);
  }
  return true;
}

}  // namespace mr

using namespace mr;

#define DISPATCH_T(dtype, ...)                                   \
  if ((dtype) == MR_F32) { typedef float T; __VA_ARGS__; }       \
  else if ((dtype) == MR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { mr::set_error("bad dtype %d", (dtype)); return MR_ERR_DTYPE; }

extern "C" {

// [GEMM(context) + GRU gates] of one decode step (see the kernels above).  ctx [M, K] (ld ldc), w_ic [3H, K] (row g*H + j =
// gate g of unit j), G = word table [classes, >= 3H] gathered by idx (null: row m), gh [M, >= 3H] with leading dimension ldgh,
// h / hnew [M, H] contiguous, save f32 [M, 3H].  M <= 32, K a multiple of the 16-byte vector.
int mr_gemm_gru_fwd(int dtype, const void* ctx, long long ldc, const void* w_ic, long long ldw, const void* G, long long ldG,
                    const long long* idx, const void* gh, long long ldgh, const void* h, void* hnew, float* save, int M, int H,
                    int K, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(M >= 1 && M <= 32 && H > 0 && K > 0 && K % vec == 0 && ldc >= K && ldw >= K && ldG >= 3 * H && ldgh >= 3 * H,
               "mr_gemm_gru_fwd: bad shape M=%d H=%d K=%d", M, H, K);
  DISPATCH_T(dtype, hipLaunchKernelGGL((gemm_nt_skinny_gru_fwd_kernel<T>), dim3(cdiv(H, 16)), dim3(256), 0, stream,
                                       (const T*)ctx, ldc, (const T*)w_ic, ldw, (const T*)G, ldG, idx, (const T*)gh, ldgh,
                                       (const T*)h, (T*)hnew, save, M, H, K));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// [GEMM + GRU backward]: dh_a = dhc [M, K] * w_t [H, K]^T, then mr_gru_bwd2 with that dh_a (never stored).  dh_b / dh_c
// nullable; dh_prev may alias dh_b (every element is read and written by one thread).
int mr_gemm_gru_bwd(int dtype, const void* dhc, long long lda, const void* w_t, long long ldw, const void* dh_b,
                    const void* dh_c, const float* save, const void* gh, long long ldgh, const void* h, void* dgi, void* dgh,
                    long long lddgh, void* dh_prev, int M, int H, int K, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(M >= 1 && M <= 32 && H > 0 && K > 0 && K % vec == 0 && lda >= K && ldw >= K && ldgh >= 3 * H && lddgh >= 3 * H,
               "mr_gemm_gru_bwd: bad shape M=%d H=%d K=%d", M, H, K);
  DISPATCH_T(dtype, hipLaunchKernelGGL((gemm_nt_skinny_gru_bwd_kernel<T>), dim3(cdiv(H, 16)), dim3(256), 0, stream,
                                       (const T*)dhc, lda, (const T*)w_t, ldw, (const T*)dh_b, (const T*)dh_c, save,
                                       (const T*)gh, ldgh, (const T*)h, (T*)dgi, (T*)dgh, lddgh, (T*)dh_prev, M, H, K));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// [out + NLL]: logits = h W^T + b in f32, then mr_nll_step_feed_fwd on them (feed_flag / feed_idx nullable: no feedback
// written).  h [N, K] (ld ldh), W [C, K] (ld ldw), C <= 256, K a multiple of the 16-byte vector.
int mr_out_nll_fwd(int dtype, const void* h, long long ldh, const void* W, long long ldw, const float* bias,
                   const long long* target, long long tstride, const float* mask, float* lp, float* loss, long long* argmax,
                   const int* feed_flag, long long* feed_idx, int N, int C, int K, int accumulate, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(N > 0 && C > 0 && C <= 256 && K > 0 && K % vec == 0 && ldh >= K && ldw >= K && target != nullptr && lp != nullptr,
               "mr_out_nll_fwd: bad shape N=%d C=%d K=%d", N, C, K);
  DISPATCH_T(dtype, hipLaunchKernelGGL((out_nll_fwd_kernel<T>), dim3(N), dim3(256), 0, stream, (const T*)h, ldh, (const T*)W,
                                       ldw, bias, target, tstride, mask, lp, loss, argmax, feed_flag, feed_idx, C, K,
                                       accumulate));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
