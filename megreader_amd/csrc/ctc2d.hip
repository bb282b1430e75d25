// 2D-CTC loss (arXiv 1907.09705) forward (alpha, nll) and backward (beta, gradient) for gfx950.
// Replaces the reference's CUDA extension ops/ctc_2d (csrc/cuda/ctc2d_cuda_kernel.cu:55-211, 254-368, 427-517;
// SURVEY.md Appendix A.1).  Semantics reproduced exactly, including the gradient convention of the "collect"
// kernel: classes that collect no finite alpha*beta mass get 0, all others (exp(lp) - exp(G + nll - lp)) * grad_out.
//
// Design (not a port): the reference maps one thread to a (batch, state) pair, packs 15 samples into 18
// workgroups for the whole batch and recomputes the height log-sum-exp three times per state and step.  Here one
// workgroup owns one sample; per step the height sums A[s] = LSE_h alpha[t-1,h,s] are computed once and kept in
// LDS, the three-way transition lse3 once per state, and the H additions of lp[t,h,l'_s] are fused with the next
// step's height reduction.  The backward kernel runs the beta recursion the same way, then computes the
// gradient for all (t,h) in parallel with one thread per (t, h, target-class) "owner" -- no atomics, no races
// (the reference's collect kernel races when T does not divide 1024).
// The work is a 32-step dependency chain on ~64 MB of traffic per step at the benchmark size: latency-bound,
// not HBM-bound; 256 workgroups (one per sample) cover the 256 CUs.
#include "common.h"
#include "../../include/megreader_hip.h"

namespace mr {

__device__ __forceinline__ float lse2f(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3f(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

template <typename T>
__global__ __launch_bounds__(128) void ctc2d_fwd_kernel(const T* __restrict__ lp, const long long* __restrict__ targets,
                                                        const long long* __restrict__ in_len,
                                                        const long long* __restrict__ tg_len, int Tn, int H, int N,
                                                        int C, int S, int blank, float* __restrict__ alpha,
                                                        float* __restrict__ nll_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPm = 2 * S + 1;
  float* A = (float*)smem_raw;   // [SPm] height log-sum of the previous step
  float* tr = A + SPm;           // [SPm] transition term of the current step
  int* lab = (int*)(tr + SPm);   // [SPm]

  const int b = blockIdx.x, tid = threadIdx.x;
  int Tb = (int)in_len[b];
  if (Tb > Tn) Tb = Tn;
  int L = (int)tg_len[b];
  if (L > S) L = S;
  const int SP = 2 * L + 1;
  for (int s = tid; s < SPm; s += 128)
    lab[s] = (s < SP && (s & 1)) ? (int)targets[(long long)b * S + (s >> 1)] : blank;
  __syncthreads();

  float* ab = alpha + (long long)b * Tn * H * SPm;
  const long long hs = (long long)N * C;          // lp stride between heights
  const long long ts = (long long)H * N * C;      // lp stride between time steps
  const T* lpb = lp + (long long)b * C;

  // t = 0
  for (int s = tid; s < SPm; s += 128) {
    float mx = -INFINITY;
    const bool on = (s == 0) || (s == 1 && L > 0);
    if (on)
      for (int h = 0; h < H; ++h) mx = fmaxf(mx, to_f32(lpb[h * hs + lab[s]]));
    float sum = 0.f;
    for (int h = 0; h < H; ++h) {
      const float v = on ? to_f32(lpb[h * hs + lab[s]]) : -INFINITY;
      ab[(long long)h * SPm + s] = v;
      if (on && mx != -INFINITY) sum += expf(v - mx);
    }
    A[s] = (on && mx != -INFINITY) ? mx + logf(sum) : -INFINITY;
  }
  for (int t = 1; t < Tn; ++t) {
    __syncthreads();
    const bool valid_t = (t < Tb) && (L > 0);
    for (int s = tid; s < SPm; s += 128) {
      float v = -INFINITY;
      if (valid_t && s < SP) {
        const float a1 = s > 0 ? A[s - 1] : -INFINITY;
        const float a2 = (s > 1 && lab[s] != lab[s - 2]) ? A[s - 2] : -INFINITY;
        v = lse3f(A[s], a1, a2);
      }
      tr[s] = v;
    }
    __syncthreads();
    float* at = ab + (long long)t * H * SPm;
    const T* lpt = lpb + t * ts;
    for (int s = tid; s < SPm; s += 128) {
      const float trv = tr[s];
      if (valid_t && s < SP) {
        float mx = -INFINITY;
        if (trv != -INFINITY)
          for (int h = 0; h < H; ++h) mx = fmaxf(mx, trv + to_f32(lpt[h * hs + lab[s]]));
        float sum = 0.f;
        for (int h = 0; h < H; ++h) {
          const float v = trv != -INFINITY ? trv + to_f32(lpt[h * hs + lab[s]]) : -INFINITY;
          at[(long long)h * SPm + s] = v;
          if (mx != -INFINITY) sum += expf(v - mx);
        }
        A[s] = mx != -INFINITY ? mx + logf(sum) : -INFINITY;
      } else {
        for (int h = 0; h < H; ++h) at[(long long)h * SPm + s] = -INFINITY;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    float nll = INFINITY;
    if (Tb >= 1 && (L > 0 || Tb == 1)) {
      const float l1 = A[2 * L];
      const float l2 = L > 0 ? A[2 * L - 1] : -INFINITY;
      nll = -lse2f(l1, l2);
    }
    nll_out[b] = nll;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ctc2d_bwd_kernel(const float* __restrict__ grad_out, const T* __restrict__ lp,
                                                        const long long* __restrict__ targets,
                                                        const long long* __restrict__ in_len,
                                                        const long long* __restrict__ tg_len,
                                                        const float* __restrict__ nll_in,
                                                        const float* __restrict__ alpha, float* __restrict__ beta,
                                                        T* __restrict__ grad, int Tn, int H, int N, int C, int S,
                                                        int blank) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPm = 2 * S + 1;
  float* Bn = (float*)smem_raw;  // [SPm] height log-sum of beta at t+1
  float* tr = Bn + SPm;          // [SPm]
  int* lab = (int*)(tr + SPm);   // [SPm]

  const int b = blockIdx.x, tid = threadIdx.x;
  int Tb = (int)in_len[b];
  if (Tb > Tn) Tb = Tn;
  int L = (int)tg_len[b];
  if (L > S) L = S;
  const int SP = 2 * L + 1;
  for (int s = tid; s < SPm; s += 256)
    lab[s] = (s < SP && (s & 1)) ? (int)targets[(long long)b * S + (s >> 1)] : blank;

  const long long hs = (long long)N * C, ts = (long long)H * N * C;
  const T* lpb = lp + (long long)b * C;
  T* gb = grad + (long long)b * C;
  // zero this sample's gradient slice [T, H, C]
  for (int i = tid; i < Tn * H * C; i += 256) {
    const int c = i % C;
    const int th = i / C;
    gb[(long long)th * hs + c] = from_f32<T>(0.f);
  }
  __syncthreads();
  if (L == 0 || Tb < 1) return;  // degenerate targets: the reference reads out of range here; gradient stays 0

  const float* ab = alpha + (long long)b * Tn * H * SPm;
  float* bb = beta + (long long)b * Tn * H * SPm;

  // beta at t = Tb-1
  {
    const int t = Tb - 1;
    float* bt = bb + (long long)t * H * SPm;
    const T* lpt = lpb + t * ts;
    for (int s = tid; s < SPm; s += 256) {
      const bool on = (s == 2 * L) || (s == 2 * L - 1);
      float mx = -INFINITY;
      if (on)
        for (int h = 0; h < H; ++h) mx = fmaxf(mx, to_f32(lpt[h * hs + lab[s]]));
      float sum = 0.f;
      for (int h = 0; h < H; ++h) {
        const float v = on ? to_f32(lpt[h * hs + lab[s]]) : -INFINITY;
        bt[(long long)h * SPm + s] = v;
        if (on && mx != -INFINITY) sum += expf(v - mx);
      }
      Bn[s] = (on && mx != -INFINITY) ? mx + logf(sum) : -INFINITY;
    }
  }
  for (int t = Tb - 2; t >= 0; --t) {
    __syncthreads();
    for (int s = tid; s < SPm; s += 256) {
      float v = -INFINITY;
      if (s < SP) {
        const float b1 = s < 2 * L ? Bn[s + 1] : -INFINITY;
        const float b2 = (s < 2 * L - 1 && lab[s + 2] != lab[s]) ? Bn[s + 2] : -INFINITY;
        v = lse3f(Bn[s], b1, b2);
      }
      tr[s] = v;
    }
    __syncthreads();
    float* bt = bb + (long long)t * H * SPm;
    const T* lpt = lpb + t * ts;
    for (int s = tid; s < SPm; s += 256) {
      const float trv = tr[s];
      float mx = -INFINITY;
      if (s < SP && trv != -INFINITY)
        for (int h = 0; h < H; ++h) mx = fmaxf(mx, trv + to_f32(lpt[h * hs + lab[s]]));
      float sum = 0.f;
      for (int h = 0; h < H; ++h) {
        const float v = (s < SP && trv != -INFINITY) ? trv + to_f32(lpt[h * hs + lab[s]]) : -INFINITY;
        bt[(long long)h * SPm + s] = v;
        if (mx != -INFINITY) sum += expf(v - mx);
      }
      Bn[s] = mx != -INFINITY ? mx + logf(sum) : -INFINITY;
    }
  }
  __syncthreads();  // beta (global scratch, written by this workgroup) is complete

  // gradient: one work item per (t, h, entry); entry 0 = blank (all even states), entry e >= 1 = label e-1
  // (state 2e-1) if it is the first occurrence of that label, which then also collects the later occurrences.
  const float nll = nll_in[b];
  const float go = grad_out[b];
  const int entries = L + 1;
  const int items = Tb * H * entries;
  for (int it = tid; it < items; it += 256) {
    const int e = it % entries;
    const int th = it / entries;
    const int h = th % H, t = th / H;
    const float* arow = ab + ((long long)t * H + h) * SPm;
    const float* brow = bb + ((long long)t * H + h) * SPm;
    int cls;
    float G = -INFINITY;
    if (e == 0) {
      cls = blank;
      for (int s = 0; s < SP; s += 2) G = lse2f(G, arow[s] + brow[s]);
    } else {
      const int s0 = 2 * e - 1;
      cls = lab[s0];
      bool first = true;
      for (int s = 1; s < s0; s += 2)
        if (lab[s] == cls) { first = false; break; }
      if (!first) continue;
      for (int s = s0; s < SP; s += 2)
        if (lab[s] == cls) G = lse2f(G, arow[s] + brow[s]);
    }
    if (G == -INFINITY) continue;
    const long long off = (long long)t * ts + (long long)h * hs + cls;
    const float x = to_f32(lpb[off]);
    gb[off] = from_f32<T>((expf(x) - expf(G + nll - x)) * go);
  }
}

}  // namespace mr

using namespace mr;

extern "C" {

// log_probs [T,H,N,C] contiguous; targets [N,S] i64; lengths [N] i64; outputs nll f32[N], alpha f32[N,T,H,2S+1]
int mr_ctc2d_fwd(int dtype, const void* log_probs, const long long* targets, const long long* input_lengths,
                 const long long* target_lengths, int T, int H, int N, int C, int S, int blank, float* nll,
                 float* alpha, hipStream_t stream) {
  MR_CHECK_ARG(T > 0 && H > 0 && N > 0 && C > 0 && S >= 0, "mr_ctc2d_fwd: bad shape T=%d H=%d N=%d C=%d S=%d", T, H,
               N, C, S);
  MR_CHECK_ARG(blank >= 0 && blank < C, "mr_ctc2d_fwd: blank must be in label range");
  MR_CHECK_ARG(2 * S + 1 <= 8192, "mr_ctc2d_fwd: target too long");
  const size_t smem = (size_t)(2 * S + 1) * 12 + 16;
  if (dtype == MR_F32)
    hipLaunchKernelGGL((ctc2d_fwd_kernel<float>), dim3(N), dim3(128), smem, stream, (const float*)log_probs, targets,
                       input_lengths, target_lengths, T, H, N, C, S, blank, alpha, nll);
  else if (dtype == MR_BF16)
    hipLaunchKernelGGL((ctc2d_fwd_kernel<bf16_t>), dim3(N), dim3(128), smem, stream, (const bf16_t*)log_probs,
                       targets, input_lengths, target_lengths, T, H, N, C, S, blank, alpha, nll);
  else { mr::set_error("mr_ctc2d_fwd: bad dtype %d", dtype); return MR_ERR_DTYPE; }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// beta: scratch f32[N,T,H,2S+1]; grad: [T,H,N,C] of `dtype` (fully written)
int mr_ctc2d_bwd(int dtype, const float* grad_out, const void* log_probs, const long long* targets,
                 const long long* input_lengths, const long long* target_lengths, const float* nll,
                 const float* alpha, float* beta, void* grad, int T, int H, int N, int C, int S, int blank,
                 hipStream_t stream) {
  MR_CHECK_ARG(T > 0 && H > 0 && N > 0 && C > 0 && S >= 0, "mr_ctc2d_bwd: bad shape");
  MR_CHECK_ARG(blank >= 0 && blank < C, "mr_ctc2d_bwd: blank must be in label range");
  const size_t smem = (size_t)(2 * S + 1) * 12 + 16;
  if (dtype == MR_F32)
    hipLaunchKernelGGL((ctc2d_bwd_kernel<float>), dim3(N), dim3(256), smem, stream, grad_out, (const float*)log_probs,
                       targets, input_lengths, target_lengths, nll, alpha, beta, (float*)grad, T, H, N, C, S, blank);
  else if (dtype == MR_BF16)
    hipLaunchKernelGGL((ctc2d_bwd_kernel<bf16_t>), dim3(N), dim3(256), smem, stream, grad_out,
                       (const bf16_t*)log_probs, targets, input_lengths, target_lengths, nll, alpha, beta,
                       (bf16_t*)grad, T, H, N, C, S, blank);
  else { mr::set_error("mr_ctc2d_bwd: bad dtype %d", dtype); return MR_ERR_DTYPE; }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
