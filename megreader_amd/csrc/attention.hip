// Per-step kernels of the Bahdanau-attention GRU decoder (reference decoders/attention_decoder.py:146-231).
// The GEMMs of a step (hidden/context/word projections, GRU gates, output layer) run on the MFMA NT kernel; these
// kernels fuse everything between them:
//   attn_step : energy[n,t] = v . tanh(hproj[n] + eproj[n,t]);  w = softmax_t(energy);  context[n] = sum_t w[n,t] enc[n,t]
//               (the reference materialises cat([hidden x T, enc]) [N,T,1057] and runs Linear(1057->512) on it every
//                step; splitting the Linear into its hidden and encoder halves makes eproj a once-per-sequence GEMM)
//   gru_gates : r,z,n gate algebra of nn.GRUCell (gate order r,z,n) and h' = (1-z) n + z h
//   nll_step  : log_softmax + NLLLoss(reduction='none') * mask + argmax  (attention_decoder.py:95-106)
// One workgroup per batch row; T (<= 64 positions) and the channel counts are tiny: latency-bound step kernels.
#include "common.h"
#include "../../include/megreader_hip.h"

namespace mr {

// tanh of the attention energies: 1 - 2 / (1 + 2^(x * 2/ln 2)) on the hardware exp2 / rcp units (absolute error ~2e-7; exact
// limits +-1 at +-inf).  The library tanhf is ~40 VALU instructions and a decode step evaluates T*Hd = 32768 of them per sample
// on ONE workgroup: round 6 measured attn_fwd2_kernel as bound by exactly that (prefetching every global load of the step
// changed nothing; profiles/r06_attention_tanh.txt), so both directions share this form.
__device__ __forceinline__ float att_tanh(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.885390081777927f));
}

// ---------------------------------------------------------------- attention step
template <typename T>
__global__ __launch_bounds__(256) void attn_step_fwd_kernel(const T* __restrict__ hproj, const T* __restrict__ eproj,
                                                            const float* __restrict__ v, const T* __restrict__ enc,
                                                            float* __restrict__ weights, T* __restrict__ context,
                                                            int Tn, int Hd, int Ep) {
  __shared__ float en[64];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* hp = hproj + (long long)n * Hd;
  for (int t = wave; t < Tn; t += 4) {
    const T* ep = eproj + ((long long)n * Tn + t) * Hd;
    float s = 0.f;
    for (int j = lane; j < Hd; j += 64) s += v[j] * att_tanh(to_f32(hp[j]) + to_f32(ep[j]));
    s = wave_sum(s);
    if (lane == 0) en[t] = s;
  }
  __syncthreads();
  if (wave == 0) {
    float e = lane < Tn ? en[lane] : -INFINITY;
    const float mx = wave_max(e);
    float ex = lane < Tn ? expf(e - mx) : 0.f;
    const float sm = wave_sum(ex);
    if (lane < Tn) {
      en[lane] = ex / sm;
      weights[(long long)n * Tn + lane] = ex / sm;
    }
  }
  __syncthreads();
  for (int c = tid; c < Ep; c += 256) {
    float s = 0.f;
    for (int t = 0; t < Tn; ++t) s += en[t] * to_f32(enc[((long long)n * Tn + t) * Ep + c]);
    context[(long long)n * Ep + c] = from_f32<T>(s);
  }
}

// backward of one attention step.  deproj / denc are ACCUMULATED (+=) into f32 buffers shared by all steps; dv is
// accumulated with atomics; dhproj is written.
template <typename T>
__global__ __launch_bounds__(256) void attn_step_bwd_kernel(const T* __restrict__ dcontext,
                                                            const float* __restrict__ dweights,
                                                            const T* __restrict__ hproj, const T* __restrict__ eproj,
                                                            const float* __restrict__ v, const T* __restrict__ enc,
                                                            const float* __restrict__ weights, T* __restrict__ dhproj,
                                                            float* __restrict__ deproj, float* __restrict__ dv,
                                                            float* __restrict__ denc, int Tn, int Hd, int Ep) {
  __shared__ float dw[64], de[64], wsh[64];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = tid; t < Tn; t += 256) wsh[t] = weights[(long long)n * Tn + t];
  // dw[t] = dcontext . enc[n,t]  (+ upstream gradient of the returned attention map)
  for (int t = wave; t < Tn; t += 4) {
    float s = 0.f;
    for (int c = lane; c < Ep; c += 64)
      s += to_f32(dcontext[(long long)n * Ep + c]) * to_f32(enc[((long long)n * Tn + t) * Ep + c]);
    s = wave_sum(s);
    if (lane == 0) dw[t] = s + (dweights ? dweights[(long long)n * Tn + t] : 0.f);
  }
  __syncthreads();
  if (wave == 0) {
    const float w = lane < Tn ? wsh[lane] : 0.f;
    const float d = lane < Tn ? dw[lane] : 0.f;
    const float dot = wave_sum(w * d);
    if (lane < Tn) de[lane] = w * (d - dot);
  }
  __syncthreads();
  // denc[n,t,c] += w[t] * dcontext[c]
  for (int i = tid; i < Tn * Ep; i += 256) {
    const int t = i / Ep, c = i - t * Ep;
    denc[((long long)n * Tn + t) * Ep + c] += wsh[t] * to_f32(dcontext[(long long)n * Ep + c]);
  }
  // through tanh: g[t,j] = de[t] * v[j] * (1 - th^2);  dhproj[j] = sum_t g; deproj[t,j] += g; dv[j] += de[t]*th
  for (int j = tid; j < Hd; j += 256) {
    const float hj = to_f32(hproj[(long long)n * Hd + j]);
    const float vj = v[j];
    float dh = 0.f, dvj = 0.f;
    for (int t = 0; t < Tn; ++t) {
      const long long o = ((long long)n * Tn + t) * Hd + j;
      const float th = att_tanh(hj + to_f32(eproj[o]));
      const float g = de[t] * vj * (1.f - th * th);
      dh += g;
      deproj[o] += g;
      dvj += de[t] * th;
    }
    dhproj[(long long)n * Hd + j] = from_f32<T>(dh);
    atomicAdd(dv + j, dvj);
  }
}

// ---------------------------------------------------------------- GRU gates (nn.GRUCell, gate order r, z, n)
// gi = gi_a (+ gi_b) : input projections incl. b_ih;  gh : hidden projection incl. b_hh;  all [N, 3H]
template <typename T>
__global__ void gru_gates_fwd_kernel(const T* __restrict__ gi_a, const T* __restrict__ gi_b,
                                     const T* __restrict__ gh, const T* __restrict__ h, T* __restrict__ hnew,
                                     float* __restrict__ save, int N, int H) {
  const long long total = (long long)N * H;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % H);
    const long long n = i / H;
    const long long b3 = n * 3 * H;
    float ir = to_f32(gi_a[b3 + j]), iz = to_f32(gi_a[b3 + H + j]), in_ = to_f32(gi_a[b3 + 2 * H + j]);
    if (gi_b) {
      ir += to_f32(gi_b[b3 + j]);
      iz += to_f32(gi_b[b3 + H + j]);
      in_ += to_f32(gi_b[b3 + 2 * H + j]);
    }
    const float hr = to_f32(gh[b3 + j]), hz = to_f32(gh[b3 + H + j]), hn = to_f32(gh[b3 + 2 * H + j]);
    const float r = sigmoidf_(ir + hr), z = sigmoidf_(iz + hz);
    const float nn_ = tanhf_(in_ + r * hn);
    const float hp = to_f32(h[i]);
    hnew[i] = from_f32<T>((1.f - z) * nn_ + z * hp);
    save[b3 + j] = r;
    save[b3 + H + j] = z;
    save[b3 + 2 * H + j] = nn_;
  }
}

// given dh' : dgi [N,3H] (same for gi_a and gi_b), dgh [N,3H], dh_prev [N,H]
template <typename T>
__global__ void gru_gates_bwd_kernel(const T* __restrict__ dhnew, const float* __restrict__ save,
                                     const T* __restrict__ gh, const T* __restrict__ h, T* __restrict__ dgi,
                                     T* __restrict__ dgh, T* __restrict__ dh, int N, int H) {
  const long long total = (long long)N * H;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % H);
    const long long n = i / H;
    const long long b3 = n * 3 * H;
    const float r = save[b3 + j], z = save[b3 + H + j], nn_ = save[b3 + 2 * H + j];
    const float hn = to_f32(gh[b3 + 2 * H + j]);
    const float hp = to_f32(h[i]);
    const float g = to_f32(dhnew[i]);
    const float dn = g * (1.f - z);
    const float dz = g * (hp - nn_);
    const float dpre_n = dn * (1.f - nn_ * nn_);
    const float dr = dpre_n * hn;
    const float dpre_r = dr * r * (1.f - r);
    const float dpre_z = dz * z * (1.f - z);
    dgi[b3 + j] = from_f32<T>(dpre_r);
    dgi[b3 + H + j] = from_f32<T>(dpre_z);
    dgi[b3 + 2 * H + j] = from_f32<T>(dpre_n);
    dgh[b3 + j] = from_f32<T>(dpre_r);
    dgh[b3 + H + j] = from_f32<T>(dpre_z);
    dgh[b3 + 2 * H + j] = from_f32<T>(dpre_n * r);
    dh[i] = from_f32<T>(g * z);
  }
}

// ---------------------------------------------------------------- log-softmax + masked NLL + argmax (one wave per row)
template <typename T>
__global__ void nll_step_fwd_kernel(const T* __restrict__ logits, int ldl, const long long* __restrict__ target,
                                    long long tstride, const float* __restrict__ mask, float* __restrict__ lp,
                                    float* __restrict__ loss, long long* __restrict__ argmax, int N, int C,
                                    int accumulate, int softmax_out, const int* __restrict__ feed_flag,
                                    long long* __restrict__ feed_idx) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const T* row = logits + (long long)n * ldl;
  float mx = -INFINITY;
  int am = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float x = to_f32(row[c]);
    if (x > mx) { mx = x; am = c; }
  }
  // wave arg-max with first-index tie break
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omx = __shfl_xor(mx, o, 64);
    const int oam = __shfl_xor(am, o, 64);
    if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
  }
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(to_f32(row[c]) - mx);
  se = wave_sum(se);
  const float lz = mx + logf(se);
  for (int c = lane; c < C; c += 64) {
    const float l = to_f32(row[c]) - lz;
    lp[(long long)n * C + c] = softmax_out ? expf(l) : l;
  }
  if (lane == 0) {
    if (argmax) argmax[n] = am;
    // word fed to the NEXT decode step: the target (teacher forcing) or this step's arg-max (attention_decoder.py:107-110),
    // chosen by a DEVICE flag so that a captured hipGraph replays the decision of the current step, not of the captured one
    if (feed_idx) feed_idx[n] = (feed_flag && *feed_flag != 0) ? target[(long long)n * tstride] : (long long)am;
    if (loss) {
      const long long tg = target[(long long)n * tstride];
      const float l = -(to_f32(row[tg]) - lz) * (mask ? mask[n] : 1.f);
      loss[n] = accumulate ? loss[n] + l : l;
    }
  }
}

// dlogits[n,c] = gloss[n] * mask[n] * (exp(lp[n,c]) - [c == target])
template <typename T>
__global__ void nll_step_bwd_kernel(const float* __restrict__ gloss, const float* __restrict__ lp,
                                    const long long* __restrict__ target, long long tstride,
                                    const float* __restrict__ mask, T* __restrict__ dlogits, int ldd, int N, int C) {
  const long long total = (long long)N * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long n = i / C;
    const float g = gloss[n] * (mask ? mask[n] : 1.f);
    const float d = expf(lp[i]) - (c == (int)target[n * tstride] ? 1.f : 0.f);
    dlogits[n * ldd + c] = from_f32<T>(g * d);
  }
}

static inline int grid_for(long long n, int block, int max_blocks = 8192) {
  long long b = (n + block - 1) / block;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

// Word embedding of the attention decoder (decoders/attention_decoder.py:187-193, nn.Embedding(V, V) initialised to
// the identity and TRAINABLE): out[n, 0:D] = table[idx[n], :] in the compute dtype, columns D..ldo-1 zero (the padded
// input of word_linear); backward scatter-adds dout rows into the f32 table gradient (several samples usually pick
// the same row: atomics).
template <typename T>
__global__ void embed_rows_fwd_kernel(const long long* __restrict__ idx, const float* __restrict__ table, T* __restrict__ out,
                                      int N, int V, int D, int ldo) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * ldo) return;
  const int n = gid / ldo, c = gid - n * ldo;
  const long long r = idx[n];
  float v = 0.f;
  if (c < D && r >= 0 && r < V) v = table[r * D + c];
  out[gid] = from_f32<T>(v);
}

template <typename T>
__global__ void embed_rows_bwd_kernel(const long long* __restrict__ idx, const T* __restrict__ dout, float* __restrict__ dtable,
                                      int N, int V, int D, int ldo) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * D) return;
  const int n = gid / D, c = gid - n * D;
  const long long r = idx[n];
  if (r >= 0 && r < V) atomicAdd(dtable + r * D + c, to_f32(dout[(long long)n * ldo + c]));
}


// =====================================================================================================================
// Round-3 decode-loop kernels (megreader_amd/decoders/attention_decoder.py:_DecodeLoopFn): the 32-step loop runs as ONE
// autograd Function with 6 launches per step forward and 6 backward (11 / ~20 + ATen adds before).  What changed for the
// kernels: operands are column slices of wider per-step buffers (leading dimensions instead of contiguous tensors: hproj and
// the hidden GRU projection come out of ONE GEMM against the stacked [W_attn_h; W_hh]), the word path is a row gather from a
// precomputed [classes, 3H] table, the backward attention kernel is parallel over (sample, 64-unit slice) instead of one
// workgroup per sample (76.8 us per launch at N = 32, profiles/r03_fpn_attention_kernel_stats_v0_baseline.csv), and the
// encoder-side gradient sum_s w_s dcontext_s is ONE kernel after the loop instead of a read-modify-write per step.
// =====================================================================================================================

// 16-byte vector of T as floats
template <typename T> struct AttVec;
template <> struct AttVec<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    const bf16_t* q = (const bf16_t*)&v;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)q[j];
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float* f) {
    const uint4 v = *(const uint4*)p;
    unpack(v, f);
  }
};
template <> struct AttVec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    const float* q = (const float*)&v;
    f[0] = q[0]; f[1] = q[1]; f[2] = q[2]; f[3] = q[3];
  }
  static __device__ __forceinline__ void load(const float* p, float* f) {
    const f32x4 v = *(const f32x4*)p;
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
  }
};

// energy / softmax / context of one decode step; hproj rows have leading dimension ldh.  512 threads per sample.
// Vector path (Hd, Ep, ldh multiples of the 16-byte vector): a wave takes a position t and its lanes 16-byte slices of the
// Hd units; the context sum runs over (position group, channel vector) pairs and is reduced through LDS.
template <typename T>
__global__ __launch_bounds__(512) void attn_fwd2_kernel(const T* __restrict__ hproj, long long ldh,
                                                        const T* __restrict__ eproj, const float* __restrict__ v,
                                                        const T* __restrict__ enc, float* __restrict__ weights,
                                                        T* __restrict__ context, int Tn, int Hd, int Ep) {
  constexpr int VEC = AttVec<T>::N;
  __shared__ float en[64];
  extern __shared__ float part[];     // [groups][Ep] partial context sums (vector path)
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* hp = hproj + (long long)n * ldh;
  const bool vec_ok = (Hd % VEC == 0) && (Ep % VEC == 0) && (ldh % VEC == 0);
  // Round 6 fast path (the published shape: Hd = 512 in bf16): the kernel is a chain of dependent memory round trips -- 8
  // score iterations per wave, then up to 10 context iterations per thread, each behind its own global load (9.5 us per decode
  // step for ~140 KB of L2-resident operands).  Here EVERY global load of the step is issued up front -- the eproj rows of the
  // wave's positions and the enc rows of the thread's (position group, channel vector) pairs do not depend on the softmax -- so
  // the step pays one round trip; same arithmetic in the same order.
  constexpr int MAXC = 10;
  if (vec_ok && Hd == 64 * VEC && Tn <= 64) {
    const int nv = Ep / VEC;
    const int groups = max(1, min(512 / nv, Tn));
    if ((Tn + groups - 1) / groups <= MAXC) {
      const int gidx = tid / nv, cv = tid - gidx * nv;
      const bool ctx_thread = tid < groups * nv;
      uint4 er[8], ev[MAXC];
      const uint4 hv = *(const uint4*)(hp + lane * VEC);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = wave + 8 * i;
        er[i] = t < Tn ? *(const uint4*)(eproj + ((long long)n * Tn + t) * Hd + lane * VEC) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int t = gidx + groups * i;
        ev[i] = (ctx_thread && t < Tn) ? *(const uint4*)(enc + ((long long)n * Tn + t) * Ep + cv * VEC) : make_uint4(0, 0, 0, 0);
      }
      float vv[VEC], a[VEC];
#pragma unroll
      for (int q = 0; q < VEC; q += 4) {
        const f32x4 v0 = *(const f32x4*)(v + lane * VEC + q);
        vv[q] = v0[0]; vv[q + 1] = v0[1]; vv[q + 2] = v0[2]; vv[q + 3] = v0[3];
      }
      AttVec<T>::unpack(hv, a);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = wave + 8 * i;
        if (t < Tn) {                      // wave-uniform
          float b[VEC];
          AttVec<T>::unpack(er[i], b);
          float s2 = 0.f;
#pragma unroll
          for (int q = 0; q < VEC; ++q) s2 += vv[q] * att_tanh(a[q] + b[q]);
          s2 = wave_sum(s2);
          if (lane == 0) en[t] = s2;
        }
      }
      __syncthreads();
      if (wave == 0) {
        float e = lane < Tn ? en[lane] : -INFINITY;
        const float mx = wave_max(e);
        float ex = lane < Tn ? expf(e - mx) : 0.f;
        const float sm = wave_sum(ex);
        if (lane < Tn) {
          en[lane] = ex / sm;
          weights[(long long)n * Tn + lane] = ex / sm;
        }
      }
      __syncthreads();
      if (ctx_thread) {
        float acc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
          const int t = gidx + groups * i;
          if (t < Tn) {
            float f[VEC];
            AttVec<T>::unpack(ev[i], f);
            const float w = en[t];
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] += w * f[q];
          }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) part[gidx * Ep + cv * VEC + q] = acc[q];
      }
      __syncthreads();
      for (int c = tid; c < Ep; c += 512) {
        float s2 = 0.f;
        for (int gi = 0; gi < groups; ++gi) s2 += part[gi * Ep + c];
        context[(long long)n * Ep + c] = from_f32<T>(s2);
      }
      return;
    }
  }
  if (vec_ok) {
    for (int t = wave; t < Tn; t += 8) {
      const T* ep = eproj + ((long long)n * Tn + t) * Hd;
      float s = 0.f;
      for (int j = lane * VEC; j < Hd; j += 64 * VEC) {
        float a[VEC], b[VEC];
        AttVec<T>::load(hp + j, a);
        AttVec<T>::load(ep + j, b);
        const f32x4 v0 = *(const f32x4*)(v + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) s += v0[q] * att_tanh(a[q] + b[q]);
        if (VEC == 8) {
          const f32x4 v1 = *(const f32x4*)(v + j + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) s += v1[q] * att_tanh(a[4 + q] + b[4 + q]);
        }
      }
      s = wave_sum(s);
      if (lane == 0) en[t] = s;
    }
  } else {
    for (int t = wave; t < Tn; t += 8) {
      const T* ep = eproj + ((long long)n * Tn + t) * Hd;
      float s = 0.f;
      for (int j = lane; j < Hd; j += 64) s += v[j] * att_tanh(to_f32(hp[j]) + to_f32(ep[j]));
      s = wave_sum(s);
      if (lane == 0) en[t] = s;
    }
  }
  __syncthreads();
  if (wave == 0) {
    float e = lane < Tn ? en[lane] : -INFINITY;
    const float mx = wave_max(e);
    float ex = lane < Tn ? expf(e - mx) : 0.f;
    const float sm = wave_sum(ex);
    if (lane < Tn) {
      en[lane] = ex / sm;
      weights[(long long)n * Tn + lane] = ex / sm;
    }
  }
  __syncthreads();
  if (vec_ok) {
    const int nv = Ep / VEC;                       // channel vectors
    const int groups = max(1, min(512 / nv, Tn));   // position groups that fit the block
    if (tid < groups * nv) {
      const int gidx = tid / nv, cv = tid - gidx * nv;
      float acc[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
      for (int t = gidx; t < Tn; t += groups) {
        float f[VEC];
        AttVec<T>::load(enc + ((long long)n * Tn + t) * Ep + cv * VEC, f);
        const float w = en[t];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] += w * f[q];
      }
#pragma unroll
      for (int q = 0; q < VEC; ++q) part[gidx * Ep + cv * VEC + q] = acc[q];
    }
    __syncthreads();
    for (int c = tid; c < Ep; c += 512) {
      float s = 0.f;
      for (int gi = 0; gi < groups; ++gi) s += part[gi * Ep + c];
      context[(long long)n * Ep + c] = from_f32<T>(s);
    }
  } else {
    for (int c = tid; c < Ep; c += 512) {
      float s = 0.f;
      for (int t = 0; t < Tn; ++t) s += en[t] * to_f32(enc[((long long)n * Tn + t) * Ep + c]);
      context[(long long)n * Ep + c] = from_f32<T>(s);
    }
  }
}

// backward of one step, grid (N, Hd / 64): every workgroup redoes the cheap part (dw = dcontext . enc, softmax backward)
// and owns a 64-unit slice of the tanh chain: dhproj (written, leading dimension lddh), deproj (+=), dv (atomics).
// The encoder-side gradient (denc) is NOT touched here: attn_denc_kernel sums it over the steps after the loop.
// dw: all positions at once -- 8 lanes per position, 16-byte vectors (one wave per position with 2-byte loads measured
// 27 us per launch at N = 32).
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd2_kernel(const T* __restrict__ dcontext,
                                                        const float* __restrict__ dweights, long long ldw,
                                                        const T* __restrict__ hproj, long long ldh,
                                                        const T* __restrict__ eproj, const float* __restrict__ v,
                                                        const T* __restrict__ enc, const float* __restrict__ weights,
                                                        T* __restrict__ dhproj, long long lddh,
                                                        float* __restrict__ deproj, float* __restrict__ dv, int Tn,
                                                        int Hd, int Ep) {
  constexpr int VEC = AttVec<T>::N;
  __shared__ float dw[64], de[64], red[4][64], redv[4][64];
  const int n = blockIdx.x, js = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Round 6 fast path (as attn_fwd2_kernel): every global load of the step -- the dcontext / enc slices of the dw dot products
  // and the eproj / deproj elements of this workgroup's 64-unit slice of the tanh chain -- is issued before anything is computed,
  // so the launch pays one memory round trip instead of ~9 + 16 dependent ones; same arithmetic in the same order.
  constexpr int MAXI = 9;
  if (Ep % VEC == 0 && Ep / VEC <= 8 * MAXI && Tn <= 64) {
    const int nv = Ep / VEC, part8 = tid & 7, tq = tid >> 3;
    const int j = js * 64 + lane;
    uint4 av[MAXI], b0[MAXI], b1[MAXI];
    float epv[16], dpv[16];
    const T* dr = dcontext + (long long)n * Ep;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int cvi = part8 + 8 * i;
      const bool okc = cvi < nv;
      av[i] = okc ? *(const uint4*)(dr + cvi * VEC) : make_uint4(0, 0, 0, 0);
      b0[i] = (okc && tq < Tn) ? *(const uint4*)(enc + ((long long)n * Tn + tq) * Ep + cvi * VEC) : make_uint4(0, 0, 0, 0);
      b1[i] = (okc && tq + 32 < Tn) ? *(const uint4*)(enc + ((long long)n * Tn + tq + 32) * Ep + cvi * VEC) : make_uint4(0, 0, 0, 0);
    }
    float hj = 0.f, vj = 0.f;
    if (j < Hd) {
      hj = to_f32(hproj[(long long)n * ldh + j]);
      vj = v[j];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = wave + 4 * i;
      const long long o = ((long long)n * Tn + t) * Hd + j;
      const bool ok = j < Hd && t < Tn;
      epv[i] = ok ? to_f32(eproj[o]) : 0.f;
      dpv[i] = ok ? deproj[o] : 0.f;
    }
    const float w_l = lane < Tn ? weights[(long long)n * Tn + lane] : 0.f;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      float a[VEC], b[VEC];
      AttVec<T>::unpack(av[i], a);
      AttVec<T>::unpack(b0[i], b);
#pragma unroll
      for (int q = 0; q < VEC; ++q) s0 += a[q] * b[q];
      AttVec<T>::unpack(b1[i], b);
#pragma unroll
      for (int q = 0; q < VEC; ++q) s1 += a[q] * b[q];
    }
    s0 += __shfl_xor(s0, 1, 64); s0 += __shfl_xor(s0, 2, 64); s0 += __shfl_xor(s0, 4, 64);
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
    if (part8 == 0) {
      if (tq < Tn) dw[tq] = s0 + (dweights ? dweights[(long long)n * ldw + tq] : 0.f);
      if (tq + 32 < Tn) dw[tq + 32] = s1 + (dweights ? dweights[(long long)n * ldw + tq + 32] : 0.f);
    }
    __syncthreads();
    if (wave == 0) {
      const float d = lane < Tn ? dw[lane] : 0.f;
      const float dot = wave_sum(w_l * d);
      if (lane < Tn) de[lane] = w_l * (d - dot);
    }
    __syncthreads();
    float dh = 0.f, dvj = 0.f;
    if (j < Hd) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t = wave + 4 * i;
        if (t < Tn) {
          const float th = att_tanh(hj + epv[i]);
          const float gg = de[t] * vj * (1.f - th * th);
          dh += gg;
          deproj[((long long)n * Tn + t) * Hd + j] = dpv[i] + gg;
          dvj += de[t] * th;
        }
      }
    }
    red[wave][lane] = dh;
    redv[wave][lane] = dvj;
    __syncthreads();
    if (wave == 0 && j < Hd) {
      dhproj[(long long)n * lddh + j] = from_f32<T>(red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
      atomicAdd(dv + j, redv[0][lane] + redv[1][lane] + redv[2][lane] + redv[3][lane]);
    }
    return;
  }
  if (Ep % VEC == 0) {
    for (int t0 = 0; t0 < Tn; t0 += 32) {
      const int t = t0 + (tid >> 3), part8 = tid & 7;
      float s = 0.f;
      if (t < Tn) {
        const T* er = enc + ((long long)n * Tn + t) * Ep;
        const T* dr = dcontext + (long long)n * Ep;
        for (int c = part8 * VEC; c < Ep; c += 8 * VEC) {
          float a[VEC], b[VEC];
          AttVec<T>::load(dr + c, a);
          AttVec<T>::load(er + c, b);
#pragma unroll
          for (int q = 0; q < VEC; ++q) s += a[q] * b[q];
        }
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (t < Tn && part8 == 0) dw[t] = s + (dweights ? dweights[(long long)n * ldw + t] : 0.f);
    }
  } else {
    for (int t = wave; t < Tn; t += 4) {
      float s = 0.f;
      for (int c = lane; c < Ep; c += 64)
        s += to_f32(dcontext[(long long)n * Ep + c]) * to_f32(enc[((long long)n * Tn + t) * Ep + c]);
      s = wave_sum(s);
      if (lane == 0) dw[t] = s + (dweights ? dweights[(long long)n * ldw + t] : 0.f);
    }
  }
  __syncthreads();
  if (wave == 0) {
    const float w = lane < Tn ? weights[(long long)n * Tn + lane] : 0.f;
    const float d = lane < Tn ? dw[lane] : 0.f;
    const float dot = wave_sum(w * d);
    if (lane < Tn) de[lane] = w * (d - dot);
  }
  __syncthreads();
  // unit j = js*64 + lane; wave w handles the positions t = w, w+4, ...
  const int j = js * 64 + lane;
  float dh = 0.f, dvj = 0.f;
  if (j < Hd) {
    const float hj = to_f32(hproj[(long long)n * ldh + j]);
    const float vj = v[j];
    for (int t = wave; t < Tn; t += 4) {
      const long long o = ((long long)n * Tn + t) * Hd + j;
      const float th = att_tanh(hj + to_f32(eproj[o]));
      const float gg = de[t] * vj * (1.f - th * th);
      dh += gg;
      deproj[o] += gg;
      dvj += de[t] * th;
    }
  }
  red[wave][lane] = dh;
  redv[wave][lane] = dvj;
  __syncthreads();
  if (wave == 0 && j < Hd) {
    dhproj[(long long)n * lddh + j] = from_f32<T>(red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
    atomicAdd(dv + j, redv[0][lane] + redv[1][lane] + redv[2][lane] + redv[3][lane]);
  }
}

// denc[n, t, c] = sum_s weights[s, n, t] * dcontext[s, n, c]   (one launch after the loop; grid (N, ceil(Ep / 256)))
template <typename T>
__global__ __launch_bounds__(256) void attn_denc_kernel(const float* __restrict__ weights, const T* __restrict__ dcontext,
                                                        T* __restrict__ denc, int S, int N, int Tn, int Ep) {
  extern __shared__ float wsh[];   // [S][Tn]
  const int n = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  for (int i = threadIdx.x; i < S * Tn; i += 256) {
    const int s = i / Tn, t = i - s * Tn;
    wsh[i] = weights[((long long)s * N + n) * Tn + t];
  }
  __syncthreads();
  if (c >= Ep) return;
  for (int t0 = 0; t0 < Tn; t0 += 32) {
    float acc[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) acc[t] = 0.f;
    for (int s = 0; s < S; ++s) {
      const float d = to_f32(dcontext[((long long)s * N + n) * Ep + c]);
#pragma unroll
      for (int t = 0; t < 32; ++t)
        if (t0 + t < Tn) acc[t] += wsh[s * Tn + t0 + t] * d;
    }
#pragma unroll
    for (int t = 0; t < 32; ++t)
      if (t0 + t < Tn) denc[((long long)n * Tn + t0 + t) * Ep + c] = from_f32<T>(acc[t]);
  }
}

// GRU gates with strided operands: gi_a row = idx ? idx[n] : n (row gather from the word table), gh with leading dimension
// ldgh (a column slice of the stacked hidden projection).
template <typename T>
__global__ void gru_fwd2_kernel(const T* __restrict__ gi_a, long long lda, const long long* __restrict__ idx,
                                const T* __restrict__ gi_b, const T* __restrict__ gh, long long ldgh,
                                const T* __restrict__ h, T* __restrict__ hnew, float* __restrict__ save, int N, int H) {
  const long long total = (long long)N * H;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % H);
    const long long n = i / H;
    const T* ga = gi_a + (idx ? idx[n] : n) * lda;
    const T* gb = gi_b ? gi_b + n * 3 * H : nullptr;
    const T* hh = gh + n * ldgh;
    float ir = to_f32(ga[j]), iz = to_f32(ga[H + j]), in_ = to_f32(ga[2 * H + j]);
    if (gb) {
      ir += to_f32(gb[j]);
      iz += to_f32(gb[H + j]);
      in_ += to_f32(gb[2 * H + j]);
    }
    const float hr = to_f32(hh[j]), hz = to_f32(hh[H + j]), hn = to_f32(hh[2 * H + j]);
    const float r = sigmoidf_(ir + hr), z = sigmoidf_(iz + hz);
    const float nn_ = tanhf_(in_ + r * hn);
    const float hp = to_f32(h[i]);
    hnew[i] = from_f32<T>((1.f - z) * nn_ + z * hp);
    const long long b3 = n * 3 * H;
    save[b3 + j] = r;
    save[b3 + H + j] = z;
    save[b3 + 2 * H + j] = nn_;
  }
}

// dh' = dh_a + dh_b + dh_c (nullable parts: the three consumers of h' -- output layer, next step's stacked projection, next
// step's z * h path -- are summed here instead of by separate add kernels); dgi [N,3H] contiguous, dgh with leading dimension
// lddgh, dh_prev [N,H] = dh' * z.
template <typename T>
__global__ void gru_bwd2_kernel(const T* __restrict__ dh_a, const T* __restrict__ dh_b, const T* __restrict__ dh_c,
                                const float* __restrict__ save, const T* __restrict__ gh, long long ldgh,
                                const T* __restrict__ h, T* __restrict__ dgi, T* __restrict__ dgh, long long lddgh,
                                T* __restrict__ dh_prev, int N, int H) {
  const long long total = (long long)N * H;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % H);
    const long long n = i / H;
    const long long b3 = n * 3 * H;
    const float r = save[b3 + j], z = save[b3 + H + j], nn_ = save[b3 + 2 * H + j];
    const float hn = to_f32(gh[n * ldgh + 2 * H + j]);
    const float hp = to_f32(h[i]);
    float g = 0.f;
    if (dh_a) g += to_f32(dh_a[i]);
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
    T* dg = dgh + n * lddgh;
    dg[j] = from_f32<T>(dpre_r);
    dg[H + j] = from_f32<T>(dpre_z);
    dg[2 * H + j] = from_f32<T>(dpre_n * r);
    dh_prev[i] = from_f32<T>(g * z);
  }
}

// dtable[idx[r], :] += rows[r, :]   (f32 atomics; R rows of D columns, row stride ldr) -- gradient of the word-table gather
template <typename T>
__global__ void rows_scatter_add_kernel(const long long* __restrict__ idx, const T* __restrict__ rows, long long ldr,
                                        float* __restrict__ dtable, int R, int V, int D) {
  const long long total = (long long)R * D;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D);
    const long long r = i / D;
    const long long t = idx[r];
    if (t >= 0 && t < V) atomicAdd(dtable + t * D + c, to_f32(rows[r * ldr + c]));
  }
}

}  // namespace mr

using namespace mr;

#define DISPATCH_T(dtype, ...)                                   \
  if ((dtype) == MR_F32) { typedef float T; __VA_ARGS__; }       \
  else if ((dtype) == MR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { mr::set_error("bad dtype %d", (dtype)); return MR_ERR_DTYPE; }

extern "C" {

int mr_attn_step_fwd(int dtype, const void* hproj, const void* eproj, const float* v, const void* enc, float* weights,
                     void* context, int N, int Tn, int Hd, int Ep, hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && Tn > 0 && Tn <= 64 && Hd > 0 && Ep > 0, "mr_attn_step_fwd: bad shape (T must be <= 64)");
  DISPATCH_T(dtype, hipLaunchKernelGGL((attn_step_fwd_kernel<T>), dim3(N), dim3(256), 0, stream, (const T*)hproj,
                                       (const T*)eproj, v, (const T*)enc, weights, (T*)context, Tn, Hd, Ep));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_attn_step_bwd(int dtype, const void* dcontext, const float* dweights, const void* hproj, const void* eproj,
                     const float* v, const void* enc, const float* weights, void* dhproj, float* deproj, float* dv,
                     float* denc, int N, int Tn, int Hd, int Ep, hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && Tn > 0 && Tn <= 64, "mr_attn_step_bwd: bad shape");
  DISPATCH_T(dtype, hipLaunchKernelGGL((attn_step_bwd_kernel<T>), dim3(N), dim3(256), 0, stream, (const T*)dcontext,
                                       dweights, (const T*)hproj, (const T*)eproj, v, (const T*)enc, weights,
                                       (T*)dhproj, deproj, dv, denc, Tn, Hd, Ep));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_gru_gates_fwd(int dtype, const void* gi_a, const void* gi_b, const void* gh, const void* h, void* hnew,
                     float* save, int N, int H, hipStream_t stream) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((gru_gates_fwd_kernel<T>), dim3(grid_for((long long)N * H, 256)), dim3(256), 0,
                                       stream, (const T*)gi_a, (const T*)gi_b, (const T*)gh, (const T*)h, (T*)hnew,
                                       save, N, H));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_gru_gates_bwd(int dtype, const void* dhnew, const float* save, const void* gh, const void* h, void* dgi,
                     void* dgh, void* dh, int N, int H, hipStream_t stream) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((gru_gates_bwd_kernel<T>), dim3(grid_for((long long)N * H, 256)), dim3(256), 0,
                                       stream, (const T*)dhnew, save, (const T*)gh, (const T*)h, (T*)dgi, (T*)dgh,
                                       (T*)dh, N, H));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// lp: f32 [N,C] log-probabilities (or probabilities when softmax_out); loss (nullable) f32 [N], accumulate adds to it;
// target / argmax are i64 (target read with element stride tstride)
int mr_nll_step_fwd(int dtype, const void* logits, int ldl, const long long* target, long long tstride,
                    const float* mask, float* lp, float* loss, long long* argmax, int N, int C, int accumulate,
                    int softmax_out, hipStream_t stream) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((nll_step_fwd_kernel<T>), dim3(cdiv(N, 4)), dim3(256), 0, stream,
                                       (const T*)logits, ldl, target, tstride, mask, lp, loss, argmax, N, C,
                                       accumulate, softmax_out, (const int*)nullptr, (long long*)nullptr));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// mr_nll_step_fwd that also writes the word index the NEXT decode step is fed with: feed_idx[n] = *feed_flag ? target[n] :
// arg-max[n] (feed_flag: one int in device memory -- the teacher-forcing coin of this step, reference
// decoders/attention_decoder.py:107-110; null = arg-max feedback)
int mr_nll_step_feed_fwd(int dtype, const void* logits, int ldl, const long long* target, long long tstride,
                         const float* mask, float* lp, float* loss, long long* argmax, const int* feed_flag,
                         long long* feed_idx, int N, int C, int accumulate, hipStream_t stream) {
  MR_CHECK_ARG(target != nullptr && feed_idx != nullptr, "mr_nll_step_feed_fwd: target / feed_idx missing");
  DISPATCH_T(dtype, hipLaunchKernelGGL((nll_step_fwd_kernel<T>), dim3(cdiv(N, 4)), dim3(256), 0, stream,
                                       (const T*)logits, ldl, target, tstride, mask, lp, loss, argmax, N, C,
                                       accumulate, 0, feed_flag, feed_idx));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_nll_step_bwd(int dtype, const float* gloss, const float* lp, const long long* target, long long tstride,
                    const float* mask, void* dlogits, int ldd, int N, int C, hipStream_t stream) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((nll_step_bwd_kernel<T>), dim3(grid_for((long long)N * C, 256)), dim3(256), 0,
                                       stream, gloss, lp, target, tstride, mask, (T*)dlogits, ldd, N, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// out: [N, ldo] (`dtype`), table f32 [V, D] row-major, idx i64 [N]
int mr_embed_rows_fwd(int dtype, const long long* idx, const float* table, void* out, int N, int V, int D, int ldo,
                      hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && V > 0 && D > 0 && ldo >= D, "mr_embed_rows_fwd: bad shape N=%d V=%d D=%d ldo=%d", N, V, D, ldo);
  DISPATCH_T(dtype, hipLaunchKernelGGL((embed_rows_fwd_kernel<T>), dim3(grid_for((long long)N * ldo, 256)), dim3(256), 0,
                                       stream, idx, table, (T*)out, N, V, D, ldo));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// dtable f32 [V, D] is ACCUMULATED into (atomics)
int mr_embed_rows_bwd(int dtype, const long long* idx, const void* dout, float* dtable, int N, int V, int D, int ldo,
                      hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && V > 0 && D > 0 && ldo >= D, "mr_embed_rows_bwd: bad shape N=%d V=%d D=%d ldo=%d", N, V, D, ldo);
  DISPATCH_T(dtype, hipLaunchKernelGGL((embed_rows_bwd_kernel<T>), dim3(grid_for((long long)N * D, 256)), dim3(256), 0,
                                       stream, idx, (const T*)dout, dtable, N, V, D, ldo));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// ---- round-3 decode-loop entry points (kernels documented above; strided operands, see include/megreader_hip.h)
int mr_attn_fwd2(int dtype, const void* hproj, long long ldh, const void* eproj, const float* v, const void* enc,
                 float* weights, void* context, int N, int Tn, int Hd, int Ep, hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && Tn > 0 && Tn <= 64 && Hd > 0 && Ep > 0 && ldh >= Hd, "mr_attn_fwd2: bad shape (T must be <= 64)");
  // dynamic LDS: [position groups][Ep] floats of partial context sums (vector path: groups = min(512 / (Ep / VEC), Tn))
  const int vec = dtype == MR_F32 ? 4 : 8;
  const int nv = Ep % vec == 0 ? Ep / vec : Ep;
  int groups = 512 / (nv > 0 ? nv : 1);
  if (groups > Tn) groups = Tn;
  if (groups < 1) groups = 1;
  const size_t lds = (size_t)groups * Ep * sizeof(float);
  MR_CHECK_ARG(lds <= 48 * 1024 && (Ep % vec != 0 || nv <= 512), "mr_attn_fwd2: Ep too large (%d)", Ep);
  MR_CHECK_ARG(((uintptr_t)v & 15) == 0, "mr_attn_fwd2: v must be 16-byte aligned");
  DISPATCH_T(dtype, hipLaunchKernelGGL((attn_fwd2_kernel<T>), dim3(N), dim3(512), lds, stream, (const T*)hproj, ldh,
                                       (const T*)eproj, v, (const T*)enc, weights, (T*)context, Tn, Hd, Ep));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_attn_bwd2(int dtype, const void* dcontext, const float* dweights, long long ldw, const void* hproj, long long ldh,
                 const void* eproj, const float* v, const void* enc, const float* weights, void* dhproj, long long lddh,
                 float* deproj, float* dv, int N, int Tn, int Hd, int Ep, hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && Tn > 0 && Tn <= 64 && ldh >= Hd && lddh >= Hd, "mr_attn_bwd2: bad shape");
  DISPATCH_T(dtype, hipLaunchKernelGGL((attn_bwd2_kernel<T>), dim3(N, cdiv(Hd, 64)), dim3(256), 0, stream,
                                       (const T*)dcontext, dweights, ldw, (const T*)hproj, ldh, (const T*)eproj, v,
                                       (const T*)enc, weights, (T*)dhproj, lddh, deproj, dv, Tn, Hd, Ep));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_attn_denc(int dtype, const float* weights, const void* dcontext, void* denc, int S, int N, int Tn, int Ep,
                 hipStream_t stream) {
  MR_CHECK_ARG(S > 0 && N > 0 && Tn > 0 && (long long)S * Tn * 4 <= 48 * 1024, "mr_attn_denc: bad shape (S*T*4 must fit 48 KB)");
  DISPATCH_T(dtype, hipLaunchKernelGGL((attn_denc_kernel<T>), dim3(N, cdiv(Ep, 256)), dim3(256), (size_t)S * Tn * 4, stream,
                                       weights, (const T*)dcontext, (T*)denc, S, N, Tn, Ep));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_gru_fwd2(int dtype, const void* gi_a, long long lda, const long long* idx, const void* gi_b, const void* gh,
                long long ldgh, const void* h, void* hnew, float* save, int N, int H, hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && H > 0 && lda >= 3 * H && ldgh >= 3 * H, "mr_gru_fwd2: bad shape");
  DISPATCH_T(dtype, hipLaunchKernelGGL((gru_fwd2_kernel<T>), dim3(grid_for((long long)N * H, 256)), dim3(256), 0, stream,
                                       (const T*)gi_a, lda, idx, (const T*)gi_b, (const T*)gh, ldgh, (const T*)h,
                                       (T*)hnew, save, N, H));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_gru_bwd2(int dtype, const void* dh_a, const void* dh_b, const void* dh_c, const float* save, const void* gh,
                long long ldgh, const void* h, void* dgi, void* dgh, long long lddgh, void* dh_prev, int N, int H,
                hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && H > 0 && ldgh >= 3 * H && lddgh >= 3 * H, "mr_gru_bwd2: bad shape");
  DISPATCH_T(dtype, hipLaunchKernelGGL((gru_bwd2_kernel<T>), dim3(grid_for((long long)N * H, 256)), dim3(256), 0, stream,
                                       (const T*)dh_a, (const T*)dh_b, (const T*)dh_c, save, (const T*)gh, ldgh,
                                       (const T*)h, (T*)dgi, (T*)dgh, lddgh, (T*)dh_prev, N, H));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_rows_scatter_add(int dtype, const long long* idx, const void* rows, long long ldr, float* dtable, int R, int V,
                        int D, hipStream_t stream) {
  MR_CHECK_ARG(R > 0 && V > 0 && D > 0 && ldr >= D, "mr_rows_scatter_add: bad shape");
  DISPATCH_T(dtype, hipLaunchKernelGGL((rows_scatter_add_kernel<T>), dim3(grid_for((long long)R * D, 256)), dim3(256), 0,
                                       stream, idx, (const T*)rows, ldr, dtable, R, V, D));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
