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

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the same recursions in the SCALED LINEAR domain (mr_tuning.ctc_linear, default).  The log-domain kernel above spends
// its 33-step dependency chain on float64 log-sum-exps (3 exp + 1 log per state and step: ~1.5 us per step, 50 us per launch at
// T = 33 -- independent of the batch size, i.e. 4 % of the 32-crops-per-GPU step).  Here a step is two adds and a multiply:
//   alpha_t[s] = (alpha_{t-1}[s] + alpha_{t-1}[s-1] + [skip] alpha_{t-1}[s-2]) * p_t(l'_s) * 2^-e,   p = exp(lp) in float64,
// e = exponent of max_s alpha_{t-1}[s] (an exact power-of-two rescale, so no rounding is added; the exponents are summed as
// integers and enter nll once, as E ln 2).  All T x (2L+1) emission probabilities are computed up front, in parallel, into an LDS
// table.  The backward variable is stored WITHOUT the emission of its own step (B'_t = the bracket above, for beta): the
// occupancy of state s at time t is then alpha_t[s] B'_t[s] / sum_s' alpha_t[s'] B'_t[s'] -- no division by p, no nll, no exp
// in the gradient kernel.  float64 throughout: nll and the gradient agree with the log-domain kernel to ~1e-13.
// Range: within one time step, states more than ~1e-308 below the largest one flush to zero (the log domain keeps them).  That
// needs log-probability gaps of ~700 between competing paths at one step -- logits hundreds apart; such a row gets occupancy 0.
// Buffers: alpha_out[n][t][s] = scaled alpha, beta_out[n][t][s] = scaled B' (same [N][T][2S+1] doubles as the log-domain pair).
__device__ __forceinline__ double pow2_neg(int e) {   // 2^-e, -1022 <= -e <= 1023
  return __longlong_as_double((long long)(1023 - e) << 52);
}
__device__ __forceinline__ int hi_word(double v) { return (int)(__double_as_longlong(v) >> 32); }
// maximum over the wavefront of a non-negative int: four DPP row rotations (every lane of a 16-lane row then holds the row's
// maximum) + one scalar read per row -- ~30 cycles instead of six dependent ds_bpermute round trips on the per-step chain
__device__ __forceinline__ int wave_max_nonneg(int v) {
#define MR_ROW_ROR_I(V, N) __builtin_amdgcn_update_dpp(0, V, 0x120 + (N), 0xf, 0xf, false)
  v = max(v, MR_ROW_ROR_I(v, 8));
  v = max(v, MR_ROW_ROR_I(v, 4));
  v = max(v, MR_ROW_ROR_I(v, 2));
  v = max(v, MR_ROW_ROR_I(v, 1));
#undef MR_ROW_ROR_I
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

template <typename T>
__global__ __launch_bounds__(256) void ctc_fwd_lin_kernel(const T* __restrict__ logits, int ldl, const void* targets,
                                                          int tg64, const void* in_len, const void* tg_len, int len64,
                                                          int Tn, int N, int C, int S, int blank,
                                                          float* __restrict__ lp_out, double* __restrict__ alpha_out,
                                                          double* __restrict__ beta_out, double* __restrict__ nll_out,
                                                          double* __restrict__ lp64_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPmax = 2 * S + 1;
  double* ptab = (double*)smem_raw;          // [Tn][SPmax] emission probabilities of the extended target
  double* al0 = ptab + (size_t)Tn * SPmax;   // [SPmax] x 4: alpha / beta double buffers
  double* al1 = al0 + SPmax;
  double* be0 = al1 + SPmax;
  double* be1 = be0 + SPmax;
  int* lab = (int*)(be1 + SPmax);            // [SPmax]
  int* wmax = lab + SPmax;                   // [2 parities][2 halves][2 waves]: hi words of the per-wave maxima

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Tb = min((int)load_idx(in_len, b, len64), Tn);
  int L = (int)load_idx(tg_len, b, len64);
  if (L > S) L = S;
  const int SP = 2 * L + 1;

  // 1) log-softmax in f32.  Small alphabets (C <= 64: the 38-class CRNN head): 8 lanes per row, 32 rows of the sample at once --
  // the row-per-wavefront loop of ctc_fwd_kernel walks 8 rows one after the other with two 6-step butterflies each; wider
  // alphabets keep it.  Same operations per element in both forms except the order of the sum of exponentials.
  if (C <= 64) {
    const int l8 = tid & 7;
    for (int t = tid >> 3; t < Tn; t += 32) {
      const T* row = logits + ((long long)t * N + b) * ldl;
      float xv[8];
      float mx = -INFINITY;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = l8 + 8 * q;
        xv[q] = c < C ? to_f32(row[c]) : -INFINITY;
        mx = fmaxf(mx, xv[q]);
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float se = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (l8 + 8 * q < C) se += expf(xv[q] - mx);
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
      const float lz = mx + logf(se);
      float* orow = lp_out + ((long long)t * N + b) * C;
      double* orow64 = lp64_out ? lp64_out + ((long long)t * N + b) * C : nullptr;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = l8 + 8 * q;
        if (c < C) {
          const float v = xv[q] - lz;
          orow[c] = v;
          if (orow64) orow64[c] = (double)v;
        }
      }
    }
  } else {
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
      if (lp64_out) {
        double* orow64 = lp64_out + ((long long)t * N + b) * C;
        for (int c = lane; c < C; c += 64) orow64[c] = (double)(to_f32(row[c]) - lz);
      }
    }
  }
  for (int s = tid; s < SPmax; s += 256) lab[s] = (s < SP) ? ext_label(targets, tg64, (long long)b * S, s, blank) : blank;
  if (tid < 8) wmax[tid] = 0;
  __syncthreads();
  // 1b) emission table: p[t][s] = exp(lp[t][l'_s]) in float64, all (t, s) in parallel
  for (int idx = tid; idx < Tn * SP; idx += 256) {
    const int t = idx / SP, s = idx - t * SP;
    ptab[(size_t)t * SPmax + s] = exp((double)lp_out[((long long)t * N + b) * C + lab[s]]);
  }
  __syncthreads();

  // 2) alpha (threads 0..127) and beta (threads 128..255), one time step per barrier
  const bool beta_half = tid >= 128;
  const int half = beta_half ? 1 : 0, hw = wave & 1;
  const int htid = tid & 127;
  double* prev = beta_half ? be0 : al0;
  double* cur = beta_half ? be1 : al1;
  double* aout = alpha_out + (long long)b * Tn * SPmax;
  double* bout = beta_out ? beta_out + (long long)b * Tn * SPmax : nullptr;
  int e_prev = 0;          // rescale exponent taken from the previous step's maximum
  long long Esum = 0;      // alpha half: sum of the exponents applied so far
  long long E_last = 0;    // ... as of step Tb - 1
  for (int i = 0; i < Tn; ++i) {
    int mhi = 0;
    const double sc = pow2_neg(e_prev);
    if (!beta_half) {
      const int t = i;
      const double* prow = ptab + (size_t)t * SPmax;
      for (int s = htid; s < SPmax; s += 128) {
        double a = 0.0;
        if (t == 0) {
          if (Tb > 0) {
            if (s == 0) a = prow[0];
            else if (s == 1 && L > 0) a = prow[1];
          }
        } else if (t < Tb && s < SP) {
          double v = prev[s];
          if (s > 0) v += prev[s - 1];
          if (s > 1 && lab[s] != lab[s - 2]) v += prev[s - 2];
          a = v * prow[s] * sc;
        }
        cur[s] = a;
        aout[(long long)t * SPmax + s] = a;
        mhi = max(mhi, hi_word(a));
      }
      if (t > 0 && t < Tb) Esum += e_prev;
      if (t == Tb - 1) E_last = Esum;
    } else if (bout && i < Tb) {
      const int t = Tb - 1 - i;
      const double* prow = ptab + (size_t)t * SPmax;
      for (int s = htid; s < SPmax; s += 128) {
        double bv = 0.0, bp = 0.0;
        if (s < SP) {
          if (i == 0) {
            if (s == SP - 1 || s == SP - 2) { bp = 1.0; bv = prow[s]; }
          } else {
            double v = prev[s];
            if (s + 1 < SP) v += prev[s + 1];
            if (s + 2 < SP && lab[s] != lab[s + 2]) v += prev[s + 2];
            bp = v * sc;
            bv = bp * prow[s];
          }
        }
        cur[s] = bv;
        bout[(long long)t * SPmax + s] = bp;
        mhi = max(mhi, hi_word(bv));
      }
    }
    // exponent of this step's maximum (non-negative doubles order like their high words) -> next step's rescale
    mhi = wave_max_nonneg(mhi);
    if (lane == 0) wmax[(i & 1) * 4 + half * 2 + hw] = mhi;
    __syncthreads();
    const int m2 = max(wmax[(i & 1) * 4 + half * 2], wmax[(i & 1) * 4 + half * 2 + 1]);
    const int be = (m2 >> 20) & 0x7ff;
    e_prev = be == 0 ? 0 : be - 1023;
    double* tmp = prev; prev = cur; cur = tmp;
  }
  if (tid == 0) {
    double nll = INFINITY;
    if (Tb > 0) {
      const double* last = aout + (long long)(Tb - 1) * SPmax;
      const double tot = last[SP - 1] + (SP > 1 ? last[SP - 2] : 0.0);
      if (tot > 0.0) nll = -(log(tot) + (double)E_last * 0.6931471805599453094);
    }
    nll_out[b] = nll;
  }
}

// gradient from the scaled pair: occupancy[t, c] = sum_{s: l'_s = c} alpha_t[s] B'_t[s] / sum_s alpha_t[s] B'_t[s]
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_lin_kernel(const float* __restrict__ lp, const double* __restrict__ alpha,
                                                           const double* __restrict__ beta,
                                                           const double* __restrict__ nll_in, const void* targets,
                                                           int tg64, const void* in_len, const void* tg_len, int len64,
                                                           const double* __restrict__ grad_out, int Tn, int N, int C,
                                                           int S, int blank, int zero_infinity, T* __restrict__ grad,
                                                           int ldg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPmax = 2 * S + 1;
  double* ab_all = (double*)smem_raw;                       // [4][SPmax] alpha * B'
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
  if (valid && !active)
    for (int c = lane; c < C; c += 64) grad[((long long)t * N + b) * ldg + c] = from_f32<T>(0.f);
  if (valid)
    for (int c = C + lane; c < ldg; c += 64) grad[((long long)t * N + b) * ldg + c] = from_f32<T>(0.f);
  const float* lrow = lp + ((long long)(valid ? t : 0) * N + b) * C;
  double total = 0.0;
  if (active) {
    const double* arow = alpha + ((long long)b * Tn + t) * SPmax;
    const double* brow = beta + ((long long)b * Tn + t) * SPmax;
    for (int s = lane; s < SPmax; s += 64) {
      const double r = (s < SP) ? arow[s] * brow[s] : 0.0;
      ab[s] = r;
      total += r;
    }
    for (int c = lane; c < C; c += 64) rowbuf[c] = (float)exp((double)lrow[c]);
  }
  total = wave_sum(total);
  __syncthreads();
  if (active && !dead && total > 0.0) {
    const double inv = 1.0 / total;
    for (int s = lane; s < SP; s += 64) {
      if (!owner[s]) continue;
      const int l = lab[s];
      double g = ab[s];
      for (int s2 = s + 2; s2 < SP; s2 += 2)
        if (lab[s2] == l) g += ab[s2];
      if (g != 0.0) rowbuf[l] -= (float)(g * inv);
    }
  }
  __syncthreads();
  if (active) {
    T* grow = grad + ((long long)t * N + b) * ldg;
    for (int c = lane; c < C; c += 64) grow[c] = from_f32<T>((float)((double)rowbuf[c] * k));
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

// LDS of ctc_fwd_lin_kernel: emission table [T][2S+1] + four state vectors (f64), labels and the 8 reduction slots (int)
static size_t ctc_lin_smem(int T, int S) {
  const size_t sp = (size_t)(2 * S + 1);
  return ((size_t)T * sp + 4 * sp) * sizeof(double) + (sp + 8) * sizeof(int) + 16;
}
// The scaled linear-domain kernels serve a problem when its emission table fits in LDS (CRNN: T = 33, S = 25..32: 17-22 KB);
// longer sequences keep the log-domain kernels.  mr_ctc_fwd and mr_ctc_bwd must see the same mr_tuning.ctc_linear.
static bool ctc_use_linear(int T, int S) { return MR_TUNE(ctc_linear) != 0 && ctc_lin_smem(T, S) <= 64 * 1024; }

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
  if (ctc_use_linear(T, S)) {   // scaled linear-domain recursion (mr_tuning.ctc_linear); mr_ctc_bwd makes the same choice
    const size_t smem_lin = ctc_lin_smem(T, S);
    if (dtype == MR_F32)
      hipLaunchKernelGGL((ctc_fwd_lin_kernel<float>), dim3(N), dim3(256), smem_lin, stream, (const float*)logits, ldl, targets,
                         targets_i64, input_lengths, target_lengths, lengths_i64, T, N, C, S, blank, log_probs, alpha, beta, nll,
                         log_probs_f64);
    else if (dtype == MR_BF16)
      hipLaunchKernelGGL((ctc_fwd_lin_kernel<bf16_t>), dim3(N), dim3(256), smem_lin, stream, (const bf16_t*)logits, ldl,
                         targets, targets_i64, input_lengths, target_lengths, lengths_i64, T, N, C, S, blank, log_probs, alpha,
                         beta, nll, log_probs_f64);
    else { mr::set_error("mr_ctc_fwd: bad dtype %d", dtype); return MR_ERR_DTYPE; }
    if (loss)
      hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(256), 0, stream, (const double*)nll, target_lengths,
                         lengths_i64, N, S, zero_infinity, loss);
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
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
  if (ctc_use_linear(T, S)) {   // alpha / beta hold the scaled linear-domain pair (see mr_ctc_fwd)
    if (dtype == MR_F32)
      hipLaunchKernelGGL((ctc_grad_lin_kernel<float>), grid, dim3(256), smem, stream, log_probs, alpha, beta, nll, targets,
                         targets_i64, input_lengths, target_lengths, lengths_i64, grad_out, T, N, C, S, blank,
                         zero_infinity, (float*)grad_logits, ldg);
    else if (dtype == MR_BF16)
      hipLaunchKernelGGL((ctc_grad_lin_kernel<bf16_t>), grid, dim3(256), smem, stream, log_probs, alpha, beta, nll, targets,
                         targets_i64, input_lengths, target_lengths, lengths_i64, grad_out, T, N, C, S, blank,
                         zero_infinity, (bf16_t*)grad_logits, ldg);
    else { mr::set_error("mr_ctc_bwd: bad dtype %d", dtype); return MR_ERR_DTYPE; }
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
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
