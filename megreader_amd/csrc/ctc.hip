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

// Forward: log-softmax (f32), then the alpha recursion on waves 0-1 and -- concurrently, when beta_out != null --
// the beta recursion on waves 2-3.  Both are T-step dependency chains of f64 log-sum-exps; running them side by
// side costs no extra latency and leaves the gradient kernel embarrassingly parallel over (t, b).
template <typename T>
__global__ __launch_bounds__(256) void ctc_fwd_kernel(const T* __restrict__ logits, int ldl, const void* targets,
                                                      int tg64, const void* in_len, const void* tg_len, int len64,
                                                      int Tn, int N, int C, int S, int blank,
                                                      float* __restrict__ lp_out, double* __restrict__ alpha_out,
                                                      double* __restrict__ beta_out, double* __restrict__ nll_out,
                                                      double* __restrict__ lp64_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPmax = 2 * S + 1;
  double* al0 = (double*)smem_raw;  // [SPmax]
  double* al1 = al0 + SPmax;        // [SPmax]
  double* be0 = al1 + SPmax;        // [SPmax]
  double* be1 = be0 + SPmax;        // [SPmax]
  int* lab = (int*)(be1 + SPmax);   // [SPmax]

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Tb = min((int)load_idx(in_len, b, len64), Tn);
  int L = (int)load_idx(tg_len, b, len64);
  if (L > S) L = S;
  const int SP = 2 * L + 1;

  // 1) log-softmax rows t = wave, wave+4, ... in f32
  for (int t = wave; t < Tn; t += 4) {
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
    // the reference hands `log_softmax(pred).to(float64)` back to its caller (decoders/crnn.py:96): written here, exactly the
    // f32 values widened, instead of by a conversion pass over lp
    if (lp64_out) {
      double* orow64 = lp64_out + ((long long)t * N + b) * C;
      for (int c = lane; c < C; c += 64) orow64[c] = (double)(to_f32(row[c]) - lz);
    }
  }
  for (int s = tid; s < SPmax; s += 256) lab[s] = (s < SP) ? ext_label(targets, tg64, (long long)b * S, s, blank) : blank;
  __syncthreads();

  // 2) alpha (threads 0..127) and beta (threads 128..255) recursions in f64, one time step per barrier
  const bool beta_half = tid >= 128;
  const int htid = tid & 127;
  double* prev = beta_half ? be0 : al0;
  double* cur = beta_half ? be1 : al1;
  double* aout = alpha_out + (long long)b * Tn * SPmax;
  double* bout = beta_out ? beta_out + (long long)b * Tn * SPmax : nullptr;
  for (int i = 0; i < Tn; ++i) {
    if (!beta_half) {
      const int t = i;
      const float* lrow = lp_out + ((long long)t * N + b) * C;
      for (int s = htid; s < SPmax; s += 128) {
        double a = -INFINITY;
        if (t == 0) {
          if (Tb > 0) {
            if (s == 0) a = (double)lrow[blank];
            else if (s == 1 && L > 0) a = (double)lrow[lab[1]];
          }
        } else if (t < Tb && s < SP) {
          const double a0 = prev[s];
          const double a1 = s > 0 ? prev[s - 1] : -INFINITY;
          const double a2 = (s > 1 && lab[s] != lab[s - 2]) ? prev[s - 2] : -INFINITY;
          const double l = lse3(a0, a1, a2);
          if (l != -INFINITY) a = l + (double)lrow[lab[s]];
        }
        cur[s] = a;
        aout[(long long)t * SPmax + s] = a;
      }
    } else if (bout && i < Tb) {
      const int t = Tb - 1 - i;
      const float* lrow = lp_out + ((long long)t * N + b) * C;
      for (int s = htid; s < SPmax; s += 128) {
        double bv = -INFINITY;
        if (s < SP) {
          if (i == 0) {
            if (s == SP - 1 || s == SP - 2) bv = (double)lrow[lab[s]];
          } else {
            const double b0 = prev[s];
            const double b1 = s + 1 < SP ? prev[s + 1] : -INFINITY;
            const double b2 = (s + 2 < SP && lab[s] != lab[s + 2]) ? prev[s + 2] : -INFINITY;
            const double l = lse3(b0, b1, b2);
            if (l != -INFINITY) bv = l + (double)lrow[lab[s]];
          }
        }
        cur[s] = bv;
        bout[(long long)t * SPmax + s] = bv;
      }
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
    if ((zero_infinity & 1) && v == INFINITY) v = 0;
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
// occupancy[t,b,c] = sum_{s: l'_s = c} exp(alpha[t,s] + beta[t,s] + nll_b - lp[t,b,c]).  With alpha AND beta stored
// by the forward kernel every (t, b) row is independent: one wavefront per row, 4 rows per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp, const double* __restrict__ alpha,
                                                       const double* __restrict__ beta,
                                                       const double* __restrict__ nll_in, const void* targets,
                                                       int tg64, const void* in_len, const void* tg_len, int len64,
                                                       const double* __restrict__ grad_out, int Tn, int N, int C,
                                                       int S, int blank, int zero_infinity, T* __restrict__ grad,
                                                       int ldg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPmax = 2 * S + 1;
  double* ab_all = (double*)smem_raw;                       // [4][SPmax] alpha+beta
  int* lab = (int*)(ab_all + 4 * SPmax);                    // [SPmax]
  int* owner = lab + SPmax;                                 // [SPmax] 1 if first state carrying its label
  float* row_all = (float*)(owner + SPmax + (SPmax & 1));   // [4][C]

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.y * 4 + wave;
  const int Tb = min((int)load_idx(in_len, b, len64), Tn);
  int L = (int)load_idx(tg_len, b, len64);
  if (L > S) L = S;
  const int SP = 2 * L + 1;
  const double nll = nll_in[b];
  // zero_infinity bit 1: per-sample losses nll_b / L_b (reference decoders/ctc_loss.py:118-122), grad_out is [N]
  const bool per_sample = (zero_infinity & 2) != 0;
  const bool dead = ((zero_infinity & 1) && nll == INFINITY) || Tb <= 0;
  const double k = dead ? 0.0 : (per_sample ? grad_out[b] : grad_out[0] / (double)N) / (double)(L < 1 ? 1 : L);
  const bool valid = t < Tn;
  const bool active = valid && t < Tb;
  double* ab = ab_all + wave * SPmax;
  float* rowbuf = row_all + wave * C;

  for (int s = tid; s < SPmax; s += 256) lab[s] = (s < SP) ? ext_label(targets, tg64, (long long)b * S, s, blank) : blank;
  __syncthreads();
  for (int s = tid; s < SPmax; s += 256) {
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
  if (valid && !active)
    for (int c = lane; c < C; c += 64) grad[((long long)t * N + b) * ldg + c] = from_f32<T>(0.f);
  // padding columns C .. ldg-1 of every row (a 38-class head stored with 40 columns): written as zeros here, so the caller
  // needs no fill pass and the padded Linear in front can take the buffer as it is
  if (valid)
    for (int c = C + lane; c < ldg; c += 64) grad[((long long)t * N + b) * ldg + c] = from_f32<T>(0.f);
  const float* lrow = lp + ((long long)(valid ? t : 0) * N + b) * C;
  if (active) {
    const double* arow = alpha + ((long long)b * Tn + t) * SPmax;
    const double* brow = beta + ((long long)b * Tn + t) * SPmax;
    for (int s = lane; s < SPmax; s += 64) ab[s] = (s < SP) ? arow[s] + brow[s] : -INFINITY;
    for (int c = lane; c < C; c += 64) rowbuf[c] = (float)exp((double)lrow[c]);
  }
  __syncthreads();
  if (active && !dead) {
    for (int s = lane; s < SP; s += 64) {
      if (!owner[s]) continue;
      const int l = lab[s];
      double g = ab[s];
      for (int s2 = s + 2; s2 < SP; s2 += 2)
        if (lab[s2] == l) g = lse2(g, ab[s2]);
      if (g != -INFINITY) rowbuf[l] -= (float)exp(g + nll - (double)lrow[l]);
    }
  }
  __syncthreads();
  if (active) {
    T* grow = grad + ((long long)t * N + b) * ldg;
    for (int c = lane; c < C; c += 64) grow[c] = from_f32<T>((float)((double)rowbuf[c] * k));
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
               int zero_infinity, float* log_probs, double* alpha, double* beta, double* nll, double* loss,
               double* log_probs_f64, hipStream_t stream) {
  MR_CHECK_ARG(T > 0 && N > 0 && C > 0 && S >= 0, "mr_ctc_fwd: bad shape T=%d N=%d C=%d S=%d", T, N, C, S);
  MR_CHECK_ARG(blank >= 0 && blank < C, "mr_ctc_fwd: blank %d out of range", blank);
  const size_t smem = (size_t)(2 * S + 1) * (4 * sizeof(double) + sizeof(int)) + 16;
  MR_CHECK_ARG(smem <= 64 * 1024, "mr_ctc_fwd: target too long for LDS (S=%d)", S);
  if (dtype == MR_F32)
    hipLaunchKernelGGL((ctc_fwd_kernel<float>), dim3(N), dim3(256), smem, stream, (const float*)logits, ldl, targets,
                       targets_i64, input_lengths, target_lengths, lengths_i64, T, N, C, S, blank, log_probs, alpha,
                       beta, nll, log_probs_f64);
  else if (dtype == MR_BF16)
    hipLaunchKernelGGL((ctc_fwd_kernel<bf16_t>), dim3(N), dim3(256), smem, stream, (const bf16_t*)logits, ldl,
                       targets, targets_i64, input_lengths, target_lengths, lengths_i64, T, N, C, S, blank,
                       log_probs, alpha, beta, nll, log_probs_f64);
  else { mr::set_error("mr_ctc_fwd: bad dtype %d", dtype); return MR_ERR_DTYPE; }
  if (loss)
    hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(256), 0, stream, (const double*)nll, target_lengths,
                       lengths_i64, N, S, zero_infinity, loss);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_ctc_bwd(int dtype, const float* log_probs, const double* alpha, const double* beta, const double* nll,
               const void* targets,
               int targets_i64, const void* input_lengths, const void* target_lengths, int lengths_i64,
               const double* grad_out, int T, int N, int C, int S, int blank, int zero_infinity, void* grad_logits,
               int ldg, hipStream_t stream) {
  MR_CHECK_ARG(T > 0 && N > 0 && C > 0 && S >= 0, "mr_ctc_bwd: bad shape");
  MR_CHECK_ARG(beta != nullptr, "mr_ctc_bwd: beta is null (mr_ctc_fwd must be given a beta buffer when a gradient is wanted)");
  const int SP = 2 * S + 1;
  const size_t smem = (size_t)SP * (4 * sizeof(double) + 2 * sizeof(int)) + 8 + (size_t)4 * C * sizeof(float) + 16;
  MR_CHECK_ARG(smem <= 64 * 1024, "mr_ctc_bwd: alphabet/target too large for LDS (C=%d S=%d)", C, S);
  const dim3 grid(N, cdiv(T, 4));
  if (dtype == MR_F32)
    hipLaunchKernelGGL((ctc_grad_kernel<float>), grid, dim3(256), smem, stream, log_probs, alpha, beta, nll, targets,
                       targets_i64, input_lengths, target_lengths, lengths_i64, grad_out, T, N, C, S, blank,
                       zero_infinity, (float*)grad_logits, ldg);
  else if (dtype == MR_BF16)
    hipLaunchKernelGGL((ctc_grad_kernel<bf16_t>), grid, dim3(256), smem, stream, log_probs, alpha, beta, nll, targets,
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
