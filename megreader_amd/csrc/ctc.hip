// 1-D CTC loss fused with log-softmax, forward (alpha, nll) and backward (beta, gradient w.r.t. LOGITS).
// Replaces `log_softmax(pred, dim=2).to(float64)` + `nn.CTCLoss(zero_infinity=True)` at
// reference decoders/crnn.py:96-98.  The reference runs the recursion in
// float64 on float32 log-probabilities; this kernel does the same: log-softmax in f32, alpha/beta in f64.
//
// One workgroup (128 threads) per batch sample; the 2L+1 extended-target states live in LDS and
// advance one time step per barrier.  The work is tiny (T*N*C logits); the kernel is bound by the
// T-step dependency chain, not by HBM bytes.
#include "common.h"
#include "../../include/megreader_hip.h"

namespace mr {

__device__ __forceinline__ double lse2(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const double m = fmax(a, b);
  return m + log(exp(a - m) + exp(b - m));
}
__device__ __forceinline__ double lse3(double a, double b, double c) {
  const double m = fmax(a, fmax(b, c));
  if (m == -INFINITY) return -INFINITY;
  return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}

__device__ __forceinline__ long long load_idx(const void* p, long long i, int is64) {
  return is64 ? ((const long long*)p)[i] : (long long)((const int*)p)[i];
}

// extended target l'_s
__device__ __forceinline__ int ext_label(const void* tg, int tg64, long long base, int s, int blank) {
  return (s & 1) ? (int)load_idx(tg, base + (s >> 1), tg64) : blank;
}

template <typename T>
__global__ __launch_bounds__(128) void ctc_fwd_kernel(const T* __restrict__ logits, int ldl, const void* targets,
                                                      int tg64, const void* in_len, const void* tg_len, int len64,
                                                      int Tn, int N, int C, int S, int blank,
                                                      float* __restrict__ lp_out, double* __restrict__ alpha_out,
                                                      double* __restrict__ nll_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* al0 = (double*)smem_raw;        // [2S+1]
  double* al1 = al0 + (2 * S + 1);        // [2S+1]
  int* lab = (int*)(al1 + (2 * S + 1));   // [2S+1]

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Tb = min((int)load_idx(in_len, b, len64), Tn);
  int L = (int)load_idx(tg_len, b, len64);
  if (L > S) L = S;
  const int SP = 2 * L + 1;
  const int SPmax = 2 * S + 1;

  // 1) log-softmax rows t = wave, wave+2, ... in f32
  for (int t = wave; t < Tn; t += 2) {
    const T* row = logits + ((long long)t * N + b) * ldl;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, to_f32(row[c]));
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(to_f32(row[c]) - mx);
    se = wave_sum(se);
    const float lz = mx + logf(se);
    float* orow = lp_out + ((long long)t * N + b) * C;
    for (int c = lane; c < C; c += 64) orow[c] = to_f32(row[c]) - lz;
  }
  for (int s = tid; s < SPmax; s += 128) lab[s] = (s < SP) ? ext_label(targets, tg64, (long long)b * S, s, blank) : blank;
  __syncthreads();

  // 2) alpha recursion (f64)
  double* prev = al0;
  double* cur = al1;
  double* aout = alpha_out + (long long)b * Tn * SPmax;
  for (int s = tid; s < SPmax; s += 128) {
    double a = -INFINITY;
    if (Tb > 0) {
      if (s == 0) a = (double)lp_out[(long long)b * C + blank];
      else if (s == 1 && L > 0) a = (double)lp_out[(long long)b * C + lab[1]];
    }
    prev[s] = a;
    aout[s] = a;
  }
  __syncthreads();
  for (int t = 1; t < Tn; ++t) {
    const float* lrow = lp_out + ((long long)t * N + b) * C;
    for (int s = tid; s < SPmax; s += 128) {
      double a = -INFINITY;
      if (t < Tb && s < SP) {
        const double a0 = prev[s];
        const double a1 = s > 0 ? prev[s - 1] : -INFINITY;
        const double a2 = (s > 1 && lab[s] != lab[s - 2]) ? prev[s - 2] : -INFINITY;
        const double l = lse3(a0, a1, a2);
        if (l != -INFINITY) a = l + (double)lrow[lab[s]];
      }
      cur[s] = a;
      aout[(long long)t * SPmax + s] = a;
    }
    __syncthreads();
    double* tmp = prev; prev = cur; cur = tmp;
  }
  if (tid == 0) {
    double nll = INFINITY;
    if (Tb > 0) {
      const double* last = aout + (long long)(Tb - 1) * SPmax;
      const double l1 = last[SP - 1];
      const double l2 = SP > 1 ? last[SP - 2] : -INFINITY;
      nll = -lse2(l1, l2);
    }
    nll_out[b] = nll;
  }
}

// loss = mean_b( zero_inf(nll_b) / max(L_b,1) )   (reduction='mean', zero_infinity=True)
__global__ void ctc_reduce_kernel(const double* __restrict__ nll, const void* tg_len, int len64, int N, int S,
                                  int zero_infinity, double* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0;
  for (int b = threadIdx.x; b < N; b += blockDim.x) {
    double v = nll[b];
    if (zero_infinity && v == INFINITY) v = 0;
    long long L = load_idx(tg_len, b, len64);
    if (L > S) L = S;
    if (L < 1) L = 1;
    s += v / (double)L;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = red[0] / (double)N;
}

// backward: grad_logits[t,b,c] = (softmax[t,b,c] - occupancy[t,b,c]) * grad_out / (N * max(L_b,1))
template <typename T>
__global__ __launch_bounds__(128) void ctc_bwd_kernel(const float* __restrict__ lp, const double* __restrict__ alpha,
                                                      const double* __restrict__ nll_in, const void* targets,
                                                      int tg64, const void* in_len, const void* tg_len, int len64,
                                                      const double* __restrict__ grad_out, int Tn, int N, int C,
                                                      int S, int blank, int zero_infinity, T* __restrict__ grad,
                                                      int ldg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPmax = 2 * S + 1;
  double* be0 = (double*)smem_raw;       // [SPmax]
  double* be1 = be0 + SPmax;             // [SPmax]
  double* ab = be1 + SPmax;              // [SPmax] alpha+beta
  int* lab = (int*)(ab + SPmax);         // [SPmax]
  int* owner = lab + SPmax;              // [SPmax] 1 if first state carrying its label
  float* rowbuf = (float*)(owner + SPmax + (SPmax & 1));  // [C]

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int Tb = min((int)load_idx(in_len, b, len64), Tn);
  int L = (int)load_idx(tg_len, b, len64);
  if (L > S) L = S;
  const int SP = 2 * L + 1;
  const double nll = nll_in[b];
  const bool dead = (zero_infinity && nll == INFINITY) || Tb <= 0;
  const double k = dead ? 0.0 : grad_out[0] / ((double)N * (double)(L < 1 ? 1 : L));

  for (int s = tid; s < SPmax; s += 128) lab[s] = (s < SP) ? ext_label(targets, tg64, (long long)b * S, s, blank) : blank;
  __syncthreads();
  for (int s = tid; s < SPmax; s += 128) {
    int own = 0;
    if (s < SP) {
      own = 1;
      const int l = lab[s];
      for (int s2 = (s & 1); s2 < s; s2 += 2)
        if (lab[s2] == l) { own = 0; break; }
    }
    owner[s] = own;
  }
  // rows at or beyond the input length get zero gradient
  for (int t = Tb + (tid / 64); t < Tn; t += 2)
    for (int c = tid & 63; c < C; c += 64) grad[((long long)t * N + b) * ldg + c] = from_f32<T>(0.f);
  __syncthreads();
  if (Tb <= 0) return;

  const double* arow_base = alpha + (long long)b * Tn * SPmax;
  double* next = be0;
  double* cur = be1;
  for (int t = Tb - 1; t >= 0; --t) {
    const float* lrow = lp + ((long long)t * N + b) * C;
    const double* arow = arow_base + (long long)t * SPmax;
    for (int s = tid; s < SPmax; s += 128) {
      double bv = -INFINITY;
      if (s < SP) {
        if (t == Tb - 1) {
          if (s == SP - 1 || (s == SP - 2)) bv = (double)lrow[lab[s]];
        } else {
          const double b0 = next[s];
          const double b1 = s + 1 < SP ? next[s + 1] : -INFINITY;
          const double b2 = (s + 2 < SP && lab[s] != lab[s + 2]) ? next[s + 2] : -INFINITY;
          const double l = lse3(b0, b1, b2);
          if (l != -INFINITY) bv = l + (double)lrow[lab[s]];
        }
      }
      cur[s] = bv;
      ab[s] = (s < SP) ? arow[s] + bv : -INFINITY;
    }
    for (int c = tid; c < C; c += 128) rowbuf[c] = (float)exp((double)lrow[c]);
    __syncthreads();
    for (int s = tid; s < SP; s += 128) {
      if (!owner[s] || dead) continue;
      const int l = lab[s];
      double g = ab[s];
      for (int s2 = s + 2; s2 < SP; s2 += 2)
        if (lab[s2] == l) g = lse2(g, ab[s2]);
      if (g != -INFINITY) rowbuf[l] -= (float)exp(g + nll - (double)lrow[l]);
    }
    __syncthreads();
    T* grow = grad + ((long long)t * N + b) * ldg;
    for (int c = tid; c < C; c += 128) grow[c] = from_f32<T>((float)((double)rowbuf[c] * k));
    __syncthreads();
    double* tmp = next; next = cur; cur = tmp;
  }
}

// eval head: softmax over classes of logits [T,N,C] written as [N,C,1,T] f32
// (pred.permute(1,2,0).unsqueeze(2); softmax(dim=1) at reference decoders/crnn.py:101-104)
template <typename T>
__global__ void softmax_nc1t_kernel(const T* __restrict__ logits, int ldl, float* __restrict__ out, int Tn, int N,
                                    int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // row = t*N + n
  if (row >= Tn * N) return;
  const int t = row / N, n = row - t * N;
  const T* src = logits + (long long)row * ldl;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, to_f32(src[c]));
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(to_f32(src[c]) - mx);
  se = wave_sum(se);
  const float inv = 1.f / se;
  for (int c = lane; c < C; c += 64) out[((long long)n * C + c) * Tn + t] = expf(to_f32(src[c]) - mx) * inv;
}

}  // namespace mr

using namespace mr;

extern "C" {

int mr_ctc_fwd(int dtype, const void* logits, int ldl, const void* targets, int targets_i64, const void* input_lengths,
               const void* target_lengths, int lengths_i64, int T, int N, int C, int S, int blank,
               int zero_infinity, float* log_probs, double* alpha, double* nll, double* loss,
               hipStream_t stream) {
  MR_CHECK_ARG(T > 0 && N > 0 && C > 0 && S >= 0, "mr_ctc_fwd: bad shape T=%d N=%d C=%d S=%d", T, N, C, S);
  MR_CHECK_ARG(blank >= 0 && blank < C, "mr_ctc_fwd: blank %d out of range", blank);
  const size_t smem = (size_t)(2 * S + 1) * (2 * sizeof(double) + sizeof(int)) + 16;
  if (dtype == MR_F32)
    hipLaunchKernelGGL((ctc_fwd_kernel<float>), dim3(N), dim3(128), smem, stream, (const float*)logits, ldl, targets,
                       targets_i64, input_lengths, target_lengths, lengths_i64, T, N, C, S, blank, log_probs, alpha,
                       nll);
  else if (dtype == MR_BF16)
    hipLaunchKernelGGL((ctc_fwd_kernel<bf16_t>), dim3(N), dim3(128), smem, stream, (const bf16_t*)logits, ldl,
                       targets, targets_i64, input_lengths, target_lengths, lengths_i64, T, N, C, S, blank,
                       log_probs, alpha, nll);
  else { mr::set_error("mr_ctc_fwd: bad dtype %d", dtype); return MR_ERR_DTYPE; }
  if (loss)
    hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(256), 0, stream, (const double*)nll, target_lengths,
                       lengths_i64, N, S, zero_infinity, loss);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_ctc_bwd(int dtype, const float* log_probs, const double* alpha, const double* nll, const void* targets,
               int targets_i64, const void* input_lengths, const void* target_lengths, int lengths_i64,
               const double* grad_out, int T, int N, int C, int S, int blank, int zero_infinity, void* grad_logits,
               int ldg, hipStream_t stream) {
  MR_CHECK_ARG(T > 0 && N > 0 && C > 0 && S >= 0, "mr_ctc_bwd: bad shape");
  const int SP = 2 * S + 1;
  const size_t smem = (size_t)SP * (3 * sizeof(double) + 2 * sizeof(int)) + 8 + (size_t)C * sizeof(float) + 16;
  MR_CHECK_ARG(smem <= 64 * 1024, "mr_ctc_bwd: alphabet/target too large for LDS (C=%d S=%d)", C, S);
  if (dtype == MR_F32)
    hipLaunchKernelGGL((ctc_bwd_kernel<float>), dim3(N), dim3(128), smem, stream, log_probs, alpha, nll, targets,
                       targets_i64, input_lengths, target_lengths, lengths_i64, grad_out, T, N, C, S, blank,
                       zero_infinity, (float*)grad_logits, ldg);
  else if (dtype == MR_BF16)
    hipLaunchKernelGGL((ctc_bwd_kernel<bf16_t>), dim3(N), dim3(128), smem, stream, log_probs, alpha, nll, targets,
                       targets_i64, input_lengths, target_lengths, lengths_i64, grad_out, T, N, C, S, blank,
                       zero_infinity, (bf16_t*)grad_logits, ldg);
  else { mr::set_error("mr_ctc_bwd: bad dtype %d", dtype); return MR_ERR_DTYPE; }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_softmax_nc1t(int dtype, const void* logits, int ldl, float* out, int T, int N, int C, hipStream_t stream) {
  MR_CHECK_ARG(T > 0 && N > 0 && C > 0, "mr_softmax_nc1t: bad shape");
  const int rows = T * N;
  if (dtype == MR_F32)
    hipLaunchKernelGGL((softmax_nc1t_kernel<float>), dim3(cdiv(rows, 4)), dim3(256), 0, stream,
                       (const float*)logits, ldl, out, T, N, C);
  else if (dtype == MR_BF16)
    hipLaunchKernelGGL((softmax_nc1t_kernel<bf16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, stream,
                       (const bf16_t*)logits, ldl, out, T, N, C);
  else { mr::set_error("mr_softmax_nc1t: bad dtype %d", dtype); return MR_ERR_DTYPE; }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
