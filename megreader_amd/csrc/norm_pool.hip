// BatchNorm2d (training statistics) and MaxPool2d on NHWC tensors.  HBM-bound kernels.
// Reference call sites: backbones/crnn.py:17-31,49-52 (MaxPool2d((2,2)), MaxPool2d((2,2),(2,1),(0,1)),
// BatchNorm2d without activation), backbones/resnet.py:26-30 (BatchNorm2d), resnet.py:199 (MaxPool2d 3x3 s2 p1).
#include "common.h"
#include "../../include/megreader_hip.h"


namespace mr {

// ------------------------------------------------------------------ BN statistics
// sums[0..C) += sum_p x[p,c]; sums[C..2C) += sum_p x[p,c]^2  (double atomics; f32 per-thread partials)
// grid = (ceil(C/64), row_splits), block 256 (4 waves stride rows, lane = channel)
template <typename T>
__global__ void bn_stats_kernel(const T* __restrict__ x, double* __restrict__ sums, int P, int C, long long ld,
                                int rows_per_block) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int p0 = blockIdx.y * rows_per_block;
  const int p1 = min(P, p0 + rows_per_block);
  float s = 0.f, q = 0.f;
  if (c < C)
    for (int p = p0 + wave; p < p1; p += 4) {
      const float v = to_f32(x[(long long)p * ld + c]);
      s += v;
      q += v * v;
    }
  red[0][wave][lane] = s;
  red[1][wave][lane] = q;
  __syncthreads();
  if (wave == 0 && c < C) {
    const double ds = (double)red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    const double dq = (double)red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    atomicAdd(sums + c, ds);
    atomicAdd(sums + C + c, dq);
  }
}

// Vectorised per-channel reductions (used when C/VEC <= 256): a thread owns one 16-byte channel vector and every
// (256 / (C/VEC))-th row of the block's row range, so every load is 16 B per lane and a wave reads whole 128-byte
// lines (the lane-per-channel kernels above move 2 bytes per lane per load and top out near 2 TB/s).
// MODE 0: s = sum x, q = sum x^2.   MODE 1: s = sum dy', q = sum dy' * xhat   (dy' = relu-masked dy)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_reduce_vec_kernel(const T* __restrict__ a, const T* __restrict__ xin,
                                                            const T* __restrict__ yin,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            double* __restrict__ sums, int relu, int P, int C,
                                                            int rows_per_block, int ncopy) {
  constexpr int VEC = VecOf<T>::N;
  __shared__ float red[2][256 * VEC];  // [s|q][group][C] with groups * C == 256 * VEC
  // ncopy > 1: the blocks spread their 2C f64 atomics over `ncopy` copies of the accumulator (block b -> copy b % ncopy;
  // bn_finalize_kernel adds the copies).  512 blocks x 2C atomics on 2C addresses serialise in L2: the statistics pass
  // of a 16384 x 256 map took 16.7 us for 8 MB (1.3 TB/s) with one copy.
  sums += (size_t)(blockIdx.x % ncopy) * 2 * C;
  const int cv = C / VEC;
  const int groups = 256 / cv;  // >= 1
  const int g = threadIdx.x / cv, v = threadIdx.x - g * cv;
  const int p0 = blockIdx.x * rows_per_block;
  const int p1 = min(P, p0 + rows_per_block);
  float s[VEC], q[VEC], mu[VEC], rs[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    s[j] = q[j] = 0.f;
    mu[j] = 0.f;
    rs[j] = 1.f;
  }
  if (g < groups) {
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        mu[j] = mean[v * VEC + j];
        rs[j] = rstd[v * VEC + j];
      }
    }
#pragma unroll 4
    for (int p = p0 + g; p < p1; p += groups) {
      const long long i = (long long)p * cv + v;
      const uint4 av = ((const uint4*)a)[i];
      const T* pa = (const T*)&av;
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float x = to_f32(pa[j]);
          s[j] += x;
          q[j] += x * x;
        }
      } else {
        const uint4 xv = ((const uint4*)xin)[i];
        const T* px = (const T*)&xv;
        uint4 yv = make_uint4(0, 0, 0, 0);
        if (relu) yv = ((const uint4*)yin)[i];
        const T* py = (const T*)&yv;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float gd = to_f32(pa[j]);
          if (relu && !(to_f32(py[j]) > 0.f)) gd = 0.f;
          s[j] += gd;
          q[j] += gd * (to_f32(px[j]) - mu[j]) * rs[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[0][g * C + v * VEC + j] = s[j];
      red[1][g * C + v * VEC + j] = q[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double ds = 0, dq = 0;
    for (int gg = 0; gg < groups; ++gg) {
      ds += (double)red[0][gg * C + c];
      dq += (double)red[1][gg * C + c];
    }
    atomicAdd(sums + c, ds);
    atomicAdd(sums + C + c, dq);
  }
}

// mean / rstd from sums, running-stat update (PyTorch: running = (1-mom)*running + mom*stat, unbiased var)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int ncopy, int P, int C, float eps,
                                   float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;  // nn.BatchNorm2d's step counter (one launch saved)
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int k = 0; k < ncopy; ++k) {
    s1 += sums[(size_t)k * 2 * C + c];
    s2 += sums[(size_t)k * 2 * C + C + c];
  }
  const double m = s1 / P;
  double var = s2 / P - m * m;
  if (var < 0) var = 0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = P > 1 ? var * ((double)P / (P - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// y = (x-mean)*rstd*gamma + beta (+residual) (relu)   ; vectorised, C % VEC == 0
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const T* __restrict__ residual, int relu,
                                long long P, int C) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = P * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cv) * VEC;
    uint4 a = ((const uint4*)x)[i];
    T* pa = (T*)&a;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (residual) r = ((const uint4*)residual)[i];
    const T* pr = (const T*)&r;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      float v = (to_f32(pa[j]) - mean[c]) * rstd[c] * gamma[c] + beta[c];
      if (residual) v += to_f32(pr[j]);
      if (relu) v = fmaxf(v, 0.f);
      pa[j] = from_f32<T>(v);
    }
    ((uint4*)y)[i] = a;
  }
}

// backward reductions: sums[0..C) += sum dy' ; sums[C..2C) += sum dy' * xhat   (dy' = relu-masked dy)
template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                     double* __restrict__ sums, int relu, int P, int C, int rows_per_block) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int p0 = blockIdx.y * rows_per_block;
  const int p1 = min(P, p0 + rows_per_block);
  float s = 0.f, q = 0.f;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c];
    for (int p = p0 + wave; p < p1; p += 4) {
      const long long idx = (long long)p * C + c;
      float g = to_f32(dy[idx]);
      if (relu && !(to_f32(y[idx]) > 0.f)) g = 0.f;
      s += g;
      q += g * (to_f32(x[idx]) - mu) * rs;
    }
  }
  red[0][wave][lane] = s;
  red[1][wave][lane] = q;
  __syncthreads();
  if (wave == 0 && c < C) {
    const double ds = (double)red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    const double dq = (double)red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    atomicAdd(sums + c, ds);
    atomicAdd(sums + C + c, dq);
  }
}

// backward finalize: adds the accumulator copies, emits dgamma / dbeta and the two per-channel means the apply pass
// needs as f32 (sbg[c] = mean(dy'), sbg[C + c] = mean(dy' * xhat)) -- the apply pass used to convert 16 doubles per
// 16-byte vector
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, int ncopy, long long P, int C,
                                       float* __restrict__ sbg, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int k = 0; k < ncopy; ++k) {
    s1 += sums[(size_t)k * 2 * C + c];
    s2 += sums[(size_t)k * 2 * C + C + c];
  }
  const float invP = 1.f / (float)P;
  sbg[c] = (float)s1 * invP;
  sbg[C + c] = (float)s2 * invP;
  // accumulate: dgamma / dbeta are gradient sinks (views of the optimizer's flat gradient buffer)
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
}

// dx = gamma*rstd*(dy' - mean(dy') - xhat*mean(dy'*xhat)); also writes the relu-masked dy' as residual gradient when
// dres != null.
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ sbg,
                                    T* __restrict__ dx, T* __restrict__ dres, int relu, long long P, int C) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = P * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cv) * VEC;
    uint4 g = ((const uint4*)dy)[i];
    uint4 a = ((const uint4*)x)[i];
    uint4 o = make_uint4(0, 0, 0, 0);
    if (relu) o = ((const uint4*)y)[i];
    T* pg = (T*)&g;
    const T* pa = (const T*)&a;
    const T* po = (const T*)&o;
    uint4 out;
    T* pout = (T*)&out;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      float gv = to_f32(pg[j]);
      if (relu && !(to_f32(po[j]) > 0.f)) gv = 0.f;
      pg[j] = from_f32<T>(gv);
      const float xh = (to_f32(pa[j]) - mean[c]) * rstd[c];
      const float sb = sbg[c], sg = sbg[C + c];
      pout[j] = from_f32<T>(gamma[c] * rstd[c] * (gv - sb - xh * sg));
    }
    ((uint4*)dx)[i] = out;
    if (dres) ((uint4*)dres)[i] = g;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused "finalize + apply" passes (training forward and backward), C % 64 == 0.  The separate bn_finalize /
// bn_bwd_finalize launches are 1-block kernels whose cost is pure launch latency (4.8 us each, 6 per CRNN step, 120 per
// ResNet50-PPM step = 0.58 ms).  Here every workgroup owns a slab of 64 channels (blockIdx.y) and re-derives that slab's
// statistics from the accumulator copies in its prologue (64 x 2 x ncopy doubles = 8 KB from L2), so the per-channel
// arithmetic -- the same double-precision expressions as the finalize kernels, hence bit-identical outputs -- costs no
// launch; the blockIdx.x == 0 workgroups also write what the finalize kernels wrote (saved mean / rstd, running
// statistics, dgamma / dbeta).  Rows are strided over blockIdx.x; a wave reads 8 rows x 128 contiguous bytes (bf16).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fused_kernel(
    const T* __restrict__ x, T* __restrict__ y, const double* __restrict__ sums, int ncopy, float eps, float momentum,
    float* __restrict__ save_mean, float* __restrict__ save_rstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, long long* __restrict__ num_batches_tracked, const float* __restrict__ gamma,
    const float* __restrict__ beta, const T* __restrict__ residual, int relu, int P, int C) {
  constexpr int VEC = VecOf<T>::N, LANES = 64 / VEC, ROWS = 256 / LANES;
  __shared__ float s_mean[64], s_rstd[64], s_gamma[64], s_beta[64];
  const int cbase = blockIdx.y * 64;
  if (threadIdx.x < 64) {
    const int c = cbase + threadIdx.x;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < ncopy; ++k) {
      s1 += sums[(size_t)k * 2 * C + c];
      s2 += sums[(size_t)k * 2 * C + C + c];
    }
    const double m = s1 / P;
    double var = s2 / P - m * m;
    if (var < 0) var = 0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    s_mean[threadIdx.x] = mf;
    s_rstd[threadIdx.x] = rf;
    s_gamma[threadIdx.x] = gamma[c];
    s_beta[threadIdx.x] = beta[c];
    if (blockIdx.x == 0) {
      if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
      save_mean[c] = mf;
      save_rstd[c] = rf;
      if (running_mean) {
        const double unb = P > 1 ? var * ((double)P / (P - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
  }
  __syncthreads();
  const int v = threadIdx.x % LANES, r0 = threadIdx.x / LANES;
  const int cv = C / VEC;
  float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = s_mean[v * VEC + j];
    rs[j] = s_rstd[v * VEC + j];
    ga[j] = s_gamma[v * VEC + j];
    be[j] = s_beta[v * VEC + j];
  }
  for (int p = blockIdx.x * ROWS + r0; p < P; p += gridDim.x * ROWS) {
    const long long i = (long long)p * cv + blockIdx.y * LANES + v;
    uint4 a = ((const uint4*)x)[i];
    T* pa = (T*)&a;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (residual) r = ((const uint4*)residual)[i];
    const T* pr = (const T*)&r;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float val = (to_f32(pa[j]) - mu[j]) * rs[j] * ga[j] + be[j];
      if (residual) val += to_f32(pr[j]);
      if (relu) val = fmaxf(val, 0.f);
      pa[j] = from_f32<T>(val);
    }
    ((uint4*)y)[i] = a;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const double* __restrict__ sums, int ncopy,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, T* __restrict__ dx, T* __restrict__ dres,
    int relu, int P, int C) {
  constexpr int VEC = VecOf<T>::N, LANES = 64 / VEC, ROWS = 256 / LANES;
  __shared__ float s_mean[64], s_k[64], s_rstd[64], s_sb[64], s_sg[64];
  const int cbase = blockIdx.y * 64;
  if (threadIdx.x < 64) {
    const int c = cbase + threadIdx.x;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < ncopy; ++k) {
      s1 += sums[(size_t)k * 2 * C + c];
      s2 += sums[(size_t)k * 2 * C + C + c];
    }
    const float invP = 1.f / (float)P;
    s_sb[threadIdx.x] = (float)s1 * invP;
    s_sg[threadIdx.x] = (float)s2 * invP;
    s_mean[threadIdx.x] = mean[c];
    s_rstd[threadIdx.x] = rstd[c];
    s_k[threadIdx.x] = gamma[c] * rstd[c];
    if (blockIdx.x == 0) {
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
    }
  }
  __syncthreads();
  const int v = threadIdx.x % LANES, r0 = threadIdx.x / LANES;
  const int cv = C / VEC;
  float mu[VEC], rs[VEC], kk[VEC], sb[VEC], sg[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = s_mean[v * VEC + j];
    rs[j] = s_rstd[v * VEC + j];
    kk[j] = s_k[v * VEC + j];
    sb[j] = s_sb[v * VEC + j];
    sg[j] = s_sg[v * VEC + j];
  }
  for (int p = blockIdx.x * ROWS + r0; p < P; p += gridDim.x * ROWS) {
    const long long i = (long long)p * cv + blockIdx.y * LANES + v;
    uint4 g = ((const uint4*)dy)[i];
    uint4 a = ((const uint4*)x)[i];
    uint4 o = make_uint4(0, 0, 0, 0);
    if (relu) o = ((const uint4*)y)[i];
    T* pg = (T*)&g;
    const T* pa = (const T*)&a;
    const T* po = (const T*)&o;
    uint4 out;
    T* pout = (T*)&out;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float gv = to_f32(pg[j]);
      if (relu && !(to_f32(po[j]) > 0.f)) gv = 0.f;
      pg[j] = from_f32<T>(gv);
      const float xh = (to_f32(pa[j]) - mu[j]) * rs[j];
      pout[j] = from_f32<T>(kk[j] * (gv - sb[j] - xh * sg[j]));
    }
    ((uint4*)dx)[i] = out;
    if (dres) ((uint4*)dres)[i] = g;
  }
}

// ------------------------------------------------------------------ BN backward in ONE pass (round 4)
// The two-kernel backward reads dy and x (and y under a fused ReLU) twice: once for the two per-channel reductions, once to
// form dx.  For the tensors of the small-batch workloads (FPN-attention at 32 crops, the DB detector at 2 images: 59-62
// BatchNorm layers of 9 + 11 us each, launch latency included) the whole tensor fits in the registers of ONE resident grid:
// a workgroup owns a [ROWS * R rows] x [64 channels] patch, loads it once (R x 2 uint4 per thread, ReLU mask applied in
// place), reduces it, adds its partial sums to the f64 accumulators (the same scratch and copies as the two-kernel path),
// passes a barrier among the workgroups of its 64-channel slab, reads the totals back and writes dx (and the masked dy for
// the residual branch) from the registers.  One launch and one read of every operand instead of two.
//
// Inter-workgroup protocol (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"): the data
// channel is 8-byte agent-scope atomics on both sides (f64 atomicAdd / relaxed agent loads: neither is served by a
// non-coherent cache), the arrival counter is a relaxed agent atomic issued behind s_waitcnt vmcnt(0) + __syncthreads -- the
// form the split reduction of the TN kernels uses (igemm_core.h, TnArgs.grp).  The barrier needs every workgroup of the launch
// resident at once: the host only takes this path when the grid is within the occupancy the runtime reports (bn_onepass_cap).
// The wait is bounded by a wall-clock limit (2 s: a concurrent collective kernel may hold CUs for a while); on timeout the
// workgroup POISONS its outputs with NaN instead of hanging or silently using partial sums.
template <typename T, int R>
__global__ __launch_bounds__(256) void bn_bwd_onepass_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, double* __restrict__ sums, int ncopy,
    unsigned* __restrict__ counters, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
    T* __restrict__ dx, T* __restrict__ dres, int relu, int P, int C) {
  constexpr int VEC = VecOf<T>::N, LANES = 64 / VEC, ROWS = 256 / LANES;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  __shared__ float red[2][ROWS][64];
  __shared__ float s_sb[64], s_sg[64];
  __shared__ int s_dead;
  const int tid = threadIdx.x;
  const int cbase = blockIdx.y * 64;
  const int v = tid % LANES, r0 = tid / LANES;
  const int cv = C / VEC;
  // Buffer resources sized to the tensor (the host checks P * C * sizeof(T) < 2^31): one 32-bit byte offset per access, and
  // rows beyond P read as zeros / drop their stores without a branch.  The whole offset travels in the VGPR operand: the
  // SGPR offset of a buffer instruction is excluded from the bounds check.
  const int bytes = (int)((long long)P * C * (long long)sizeof(T));
  const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc((void*)dy, (short)0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, (short)0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc((void*)(relu ? y : x), (short)0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_dx = __builtin_amdgcn_make_buffer_rsrc((void*)dx, (short)0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_dr = __builtin_amdgcn_make_buffer_rsrc((void*)(dres ? dres : dx), (short)0, bytes, 0x00020000);
  const int voff = (r0 * cv + blockIdx.y * LANES + v) * 16;          // this lane inside a batch of ROWS rows
  const int row_bytes = cv * 16;
  const int sbase = blockIdx.x * (ROWS * R) * row_bytes;              // first row of this workgroup
  float mu[VEC], rs[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = mean[cbase + v * VEC + j];
    rs[j] = rstd[cbase + v * VEC + j];
  }
  u32x4 gq[R], xq[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    gq[k] = __builtin_amdgcn_raw_buffer_load_b128(r_g, voff + sbase + k * ROWS * row_bytes, 0, 0);
    xq[k] = __builtin_amdgcn_raw_buffer_load_b128(r_x, voff + sbase + k * ROWS * row_bytes, 0, 0);
  }
  float s[VEC], q[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = q[j] = 0.f;
  // y is only needed for the ReLU mask: it passes through registers in batches of RB rows instead of staying cached
  constexpr int RB = R < 4 ? R : 4;
#pragma unroll
  for (int kb = 0; kb < R; kb += RB) {
    u32x4 yq[RB];
    if (relu) {
#pragma unroll
      for (int u = 0; u < RB; ++u)
        yq[u] = __builtin_amdgcn_raw_buffer_load_b128(r_y, voff + sbase + (kb + u) * ROWS * row_bytes, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int k = kb + u;
      T* pg = (T*)&gq[k];
      const T* pa = (const T*)&xq[k];
      const T* po = (const T*)&yq[u];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float gd = to_f32(pg[j]);
        if (relu && !(to_f32(po[j]) > 0.f)) {
          gd = 0.f;
          pg[j] = from_f32<T>(0.f);
        }
        s[j] += gd;
        q[j] += gd * (to_f32(pa[j]) - mu[j]) * rs[j];     // rows beyond P read as zeros: gd == 0
      }
      __builtin_amdgcn_sched_barrier(0);   // one row's unpacked values live at a time (the scheduler interleaved all rows: 159 VGPRs)
    }
  }
  // the patch stays PACKED across the barrier: without this the compiler keeps the f32 conversions of phase 1 alive for phase 2
  // (128 more VGPRs for 8 bf16 rows)
#pragma unroll
  for (int k = 0; k < R; ++k) asm volatile("" : "+v"(gq[k]), "+v"(xq[k]));
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[0][r0][v * VEC + j] = s[j];
    red[1][r0][v * VEC + j] = q[j];
  }
  if (tid == 0) s_dead = 0;
  __syncthreads();
  double* mine = sums + (size_t)(blockIdx.x % ncopy) * 2 * C;
  if (tid < 128) {
    const int c = tid & 63, which = tid >> 6;
    double d = 0;
#pragma unroll 8
    for (int r = 0; r < ROWS; ++r) d += (double)red[which][r][c];
    atomicAdd(mine + which * C + cbase + c, d);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* ctr = counters + blockIdx.y * 128;     // one 512-byte segment per slab: arrivals of different slabs do not share a line
  if (tid == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = gridDim.x;
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0 && wall_clock64() - t0 > 200000000ll) {   // 100 MHz constant clock: 2 s
        s_dead = 1;
        break;
      }
    }
  }
  __syncthreads();
  const bool dead = s_dead != 0;
  if (tid < 64) {
    const int c = cbase + tid;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < ncopy; ++k) {
      s1 += __hip_atomic_load(sums + (size_t)k * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s2 += __hip_atomic_load(sums + (size_t)k * 2 * C + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dead) s1 = s2 = (double)__builtin_nanf("");
    const float invP = 1.f / (float)P;
    s_sb[tid] = (float)s1 * invP;
    s_sg[tid] = (float)s2 * invP;
    if (blockIdx.x == 0) {
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
    }
  }
  __syncthreads();
  float sb[VEC], sg[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sb[j] = s_sb[v * VEC + j];
    sg[j] = s_sg[v * VEC + j];
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const T* pg = (const T*)&gq[k];
    const T* pa = (const T*)&xq[k];
    u32x4 out;
    T* pout = (T*)&out;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float xh = (to_f32(pa[j]) - mu[j]) * rs[j];
      const float kj = gamma[cbase + v * VEC + j] * rs[j];
      pout[j] = from_f32<T>(kj * (to_f32(pg[j]) - sb[j] - xh * sg[j]));
    }
    __builtin_amdgcn_raw_buffer_store_b128(out, r_dx, voff + sbase + k * ROWS * row_bytes, 0, 0);
    if (dres) __builtin_amdgcn_raw_buffer_store_b128(gq[k], r_dr, voff + sbase + k * ROWS * row_bytes, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// eval-mode BN: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta
__global__ void bn_eval_coeff_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  rstd[c] = 1.f / sqrtf(rv[c] + eps);
}

// ------------------------------------------------------------------ MaxPool
// one thread per (output pixel, 16-byte channel vector); idx = first max position (i*kw + j), PyTorch order
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx,
                                   int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                                   int Ho, int Wo) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = (long long)N * Ho * Wo * cv;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(t % cv);
    long long q = t / cv;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho);
    const int n = (int)(q / Ho);
    float best[VEC];
    unsigned char bi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    bool first = true;
    for (int i = 0; i < kh; ++i) {
      const int h = ho * sh - ph + i;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int jx = 0; jx < kw; ++jx) {
        const int w = wo * sw - pw + jx;
        if ((unsigned)w >= (unsigned)W) continue;
        const uint4 v = ((const uint4*)x)[(((long long)n * H + h) * W + w) * cv + c];
        const T* pv = (const T*)&v;
        const unsigned char code = (unsigned char)(i * kw + jx);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float f = to_f32(pv[j]);
          if (first || f > best[j] || f != f) { best[j] = f; bi[j] = code; }
        }
        first = false;
      }
    }
    uint4 o;
    T* po = (T*)&o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(best[j]);
    ((uint4*)y)[t] = o;
    unsigned char* pi = idx + t * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) pi[j] = bi[j];
  }
}

// gather form (no atomics): each input pixel sums dy of the windows whose argmax points at it
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                   const T* __restrict__ relu_y, T* __restrict__ dx, int N, int H, int W, int C,
                                   int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = (long long)N * H * W * cv;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(t % cv);
    long long q = t / cv;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H);
    const int n = (int)(q / H);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int i = 0; i < kh; ++i) {
      const int hn = h + ph - i;
      if (hn < 0) continue;
      const int ho = hn / sh;
      if (ho * sh != hn || ho >= Ho) continue;
      for (int jx = 0; jx < kw; ++jx) {
        const int wn = w + pw - jx;
        if (wn < 0) continue;
        const int wo = wn / sw;
        if (wo * sw != wn || wo >= Wo) continue;
        const long long o = (((long long)n * Ho + ho) * Wo + wo) * cv + c;
        const uint4 g = ((const uint4*)dy)[o];
        const T* pg = (const T*)&g;
        const unsigned char* pi = idx + o * VEC;
        const unsigned char code = (unsigned char)(i * kw + jx);
        if (relu_y) {
          // fused ReLU backward of the layer that produced the pool input: the value at the arg-max position IS the
          // pooled output, so the mask is read at pooled resolution (a quarter of the bytes of the full-resolution
          // ReLU output)
          const uint4 yv = ((const uint4*)relu_y)[o];
          const T* py = (const T*)&yv;
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            if (pi[j] == code && to_f32(py[j]) > 0.f) acc[j] += to_f32(pg[j]);
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            if (pi[j] == code) acc[j] += to_f32(pg[j]);
        }
      }
    }
    uint4 out;
    T* po = (T*)&out;
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(acc[j]);
    ((uint4*)dx)[t] = out;
  }
}

// Same gather with the window geometry as template constants (the CRNN / ResNet pools: 2x2 stride (2,2) and (2,1),
// 3x3 stride 2): no runtime divisions (the generic kernel spends ~40 integer instructions per element on them and ran
// at 1.7 TB/s), the window loops unrolled, one 8-byte load for the arg-max codes, one (n, h) row per blockIdx.y.
template <typename T, int KH, int KW, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool_bwd_fixed_kernel(const T* __restrict__ dy,
                                                                 const unsigned char* __restrict__ idx,
                                                                 const T* __restrict__ relu_y, T* __restrict__ dx,
                                                                 int H, int W, int cv, int ph, int pw, int Ho,
                                                                 int Wo) {
  constexpr int VEC = VecOf<T>::N;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W * cv) return;
  const int w = t / cv, c = t - w * cv;      // cv is small: one division per thread, not per window tap
  const int n = blockIdx.y / H, h = blockIdx.y - n * H;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
  for (int i = 0; i < KH; ++i) {
    const int hn = h + ph - i;
    const int ho = hn / SH;
    if (hn < 0 || ho * SH != hn || ho >= Ho) continue;
#pragma unroll
    for (int jx = 0; jx < KW; ++jx) {
      const int wn = w + pw - jx;
      const int wo = wn / SW;
      if (wn < 0 || wo * SW != wn || wo >= Wo) continue;
      const long long o = (((long long)n * Ho + ho) * Wo + wo) * cv + c;
      const uint4 g = ((const uint4*)dy)[o];
      const T* pg = (const T*)&g;
      unsigned char code[8];
      if (VEC == 8) *(uint2*)code = *(const uint2*)(idx + o * VEC);
      else *(unsigned*)code = *(const unsigned*)(idx + o * VEC);
      const unsigned char want = (unsigned char)(i * KW + jx);
      if (relu_y) {
        const uint4 yv = ((const uint4*)relu_y)[o];
        const T* py = (const T*)&yv;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (code[j] == want && to_f32(py[j]) > 0.f) acc[j] += to_f32(pg[j]);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (code[j] == want) acc[j] += to_f32(pg[j]);
      }
    }
  }
  uint4 out;
  T* po = (T*)&out;
#pragma unroll
  for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(acc[j]);
  ((uint4*)dx)[((long long)blockIdx.y * W + w) * cv + c] = out;
}

// Forward with the window geometry as template constants (the pools the models use: 2x2 stride (2,2) and (2,1), 3x3 stride 2):
// one (n, ho) output row per blockIdx.y, no runtime divisions per window tap, the KH*KW loads of a thread issued back to back,
// the arg-max codes of a channel vector stored as ONE 4- / 8-byte word (the generic kernel stores them byte by byte).  Same
// comparison rule, so values and codes are bit-identical to maxpool_fwd_kernel.
template <typename T, int KH, int KW, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool_fwd_fixed_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                 unsigned char* __restrict__ idx, int H, int W, int cv,
                                                                 int ph, int pw, int Ho, int Wo) {
  constexpr int VEC = VecOf<T>::N;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * cv) return;
  const int wo = t / cv, c = t - wo * cv;
  const int n = blockIdx.y / Ho, ho = blockIdx.y - n * Ho;
  uint4 v[KH * KW];
  bool ok[KH * KW];
#pragma unroll
  for (int i = 0; i < KH; ++i)
#pragma unroll
    for (int jx = 0; jx < KW; ++jx) {
      const int h = ho * SH - ph + i, w = wo * SW - pw + jx;
      ok[i * KW + jx] = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
      v[i * KW + jx] = ok[i * KW + jx] ? ((const uint4*)x)[(((long long)n * H + h) * W + w) * cv + c] : make_uint4(0, 0, 0, 0);
    }
  float best[VEC];
  unsigned char bi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < VEC; ++j) best[j] = -INFINITY;
  bool first = true;
#pragma unroll
  for (int k = 0; k < KH * KW; ++k) {
    if (!ok[k]) continue;
    const T* pv = (const T*)&v[k];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float f = to_f32(pv[j]);
      if (first || f > best[j] || f != f) { best[j] = f; bi[j] = (unsigned char)k; }
    }
    first = false;
  }
  uint4 o;
  T* po = (T*)&o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(best[j]);
  const long long q = (((long long)n * Ho + ho) * Wo + wo) * cv + c;
  ((uint4*)y)[q] = o;
  if (VEC == 8) *(uint2*)(idx + q * VEC) = *(const uint2*)bi;
  else *(unsigned*)(idx + q * VEC) = *(const unsigned*)bi;
}

// Backward of the 2x2 / stride 2 / no padding pool on even H and W (CRNN conv1, ResNet-free): every input pixel belongs to
// exactly one window, so the pass is organised by POOLED element -- a thread loads dy, the codes and the ReLU mask source of
// one pooled channel vector once and writes the four input vectors of its window (the value at the arg-max position, zeros at
// the other three); nothing is summed, so no conversion either.  Same result as maxpool_bwd_fixed_kernel<T, 2, 2, 2, 2>.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_2x2s2_kernel(const T* __restrict__ dy,
                                                                 const unsigned char* __restrict__ idx,
                                                                 const T* __restrict__ relu_y, T* __restrict__ dx, int H,
                                                                 int W, int cv, int Ho, int Wo) {
  constexpr int VEC = VecOf<T>::N;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * cv) return;
  const int wo = t / cv, c = t - wo * cv;
  const int n = blockIdx.y / Ho, ho = blockIdx.y - n * Ho;
  const long long o = (((long long)n * Ho + ho) * Wo + wo) * cv + c;
  const uint4 g = ((const uint4*)dy)[o];
  unsigned char code[8];
  if (VEC == 8) *(uint2*)code = *(const uint2*)(idx + o * VEC);
  else *(unsigned*)code = *(const unsigned*)(idx + o * VEC);
  const T* pg = (const T*)&g;
  T gm[VEC];
  if (relu_y) {
    const uint4 yv = ((const uint4*)relu_y)[o];
    const T* py = (const T*)&yv;
#pragma unroll
    for (int j = 0; j < VEC; ++j) gm[j] = to_f32(py[j]) > 0.f ? pg[j] : from_f32<T>(0.f);
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) gm[j] = pg[j];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint4 out;
    T* po = (T*)&out;
    // (the gather kernel computes 0.f + g and rounds back: the same bits, except that it turns -0 into +0 -- kept)
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = code[j] == k ? from_f32<T>(0.f + to_f32(gm[j])) : from_f32<T>(0.f);
    ((uint4*)dx)[(((long long)n * H + 2 * ho + (k >> 1)) * W + 2 * wo + (k & 1)) * cv + c] = out;
  }
}

// Backward of the 2x2 / stride (2, 1) / padding (0, pw) pool on even H (CRNN pools 3 and 4, reference backbones/crnn.py:52-55):
// the two input rows of a window row share their windows, so one thread serves the input pixels (2*ho, w) and (2*ho + 1, w):
// the (at most) two windows wo = w + pw - {0, 1} are loaded once -- dy, codes, ReLU mask source -- instead of once per row.
// Accumulation order per output as in maxpool_bwd_fixed_kernel<T, 2, 2, 2, 1> (jx = 0, then 1): the same bits.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_2x2s21_kernel(const T* __restrict__ dy,
                                                                  const unsigned char* __restrict__ idx,
                                                                  const T* __restrict__ relu_y, T* __restrict__ dx, int H,
                                                                  int W, int cv, int pw, int Ho, int Wo) {
  constexpr int VEC = VecOf<T>::N;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W * cv) return;
  const int w = t / cv, c = t - w * cv;
  const int n = blockIdx.y / Ho, ho = blockIdx.y - n * Ho;
  float acc[2][VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = 0.f;
#pragma unroll
  for (int jx = 0; jx < 2; ++jx) {
    const int wo = w + pw - jx;
    if (wo < 0 || wo >= Wo) continue;
    const long long o = (((long long)n * Ho + ho) * Wo + wo) * cv + c;
    const uint4 g = ((const uint4*)dy)[o];
    const T* pg = (const T*)&g;
    unsigned char code[8];
    if (VEC == 8) *(uint2*)code = *(const uint2*)(idx + o * VEC);
    else *(unsigned*)code = *(const unsigned*)(idx + o * VEC);
    uint4 yv = make_uint4(0, 0, 0, 0);
    if (relu_y) yv = ((const uint4*)relu_y)[o];
    const T* py = (const T*)&yv;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const bool live = !relu_y || to_f32(py[j]) > 0.f;
      if (live && code[j] == jx) acc[0][j] += to_f32(pg[j]);
      if (live && code[j] == 2 + jx) acc[1][j] += to_f32(pg[j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint4 out;
    T* po = (T*)&out;
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(acc[i][j]);
    ((uint4*)dx)[(((long long)n * H + 2 * ho + i) * W + w) * cv + c] = out;
  }
}

#define g_bn_fused MR_TUNE(bn_fused)   // 1: fold the finalize kernels into the apply passes (mr_tuning.bn_fused)
// blocks along the rows of a fused apply pass: ~4 row groups per block (8 until round 5), at most ~16 blocks per CU in total
static inline int bn_fused_grid_x(long long P, int rows, int slabs) {
  static int groups = 0, capw = 0;   // experiment hooks (tools/): MEGREADER_BN_GROUPS row groups per block, MEGREADER_BN_CAP blocks
  if (!groups) {
    const char* e = getenv("MEGREADER_BN_GROUPS");
    groups = e && atoi(e) > 0 ? atoi(e) : 4;    // Res50-PPM step: 11.14 / 11.11 / 11.18 / 11.30 / 11.62 ms at 2 / 4 / 8 / 16 / 32
    const char* c = getenv("MEGREADER_BN_CAP");
    capw = c && atoi(c) > 0 ? atoi(c) : 4096;
  }
  long long gx = (P + (long long)rows * groups - 1) / ((long long)rows * groups);
  const long long cap = capw / slabs > 1 ? capw / slabs : 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return (int)gx;
}

#define g_bn_onepass MR_TUNE(bn_onepass)
// Workgroups of bn_bwd_onepass_kernel that are resident at once on the current device: CUs x (reported occupancy of the
// hungrier instantiation, at most 4: 49 SGPRs, so the API's VGPR-limited answer is the admission limit of
// MI355X_MICROARCH.md "Residency and cooperative launch") minus a margin of a quarter workgroup per CU.  0 = not available.
static int bn_onepass_cap() {
  static int cap = -1;
  if (cap >= 0) return cap;
  int dev = 0, cus = 0, occ_b = 0, occ_f = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, bn_bwd_onepass_kernel<bf16_t, 8>, 256, 0) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_f, bn_bwd_onepass_kernel<float, 8>, 256, 0) != hipSuccess) {
    (void)hipGetLastError();
    return cap = 0;
  }
  int occ = occ_b < occ_f ? occ_b : occ_f;
  if (occ > 4) occ = 4;
  cap = occ >= 1 ? cus * occ - cus / 4 : 0;
  return cap;
}

static inline int grid_for(long long n, int block, int max_blocks = 16384) {
  long long b = (n + block - 1) / block;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace mr

using namespace mr;

#define DISPATCH_T(dtype, ...)                                   \
  if ((dtype) == MR_F32) { typedef float T; __VA_ARGS__; }       \
  else if ((dtype) == MR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { mr::set_error("bad dtype %d", (dtype)); return MR_ERR_DTYPE; }

static int split_rows(int P, int C, int& rpb) {
  const int colg = cdiv(C, 64);
  int splits = 2048 / colg;
  if (splits < 1) splits = 1;
  if (splits > cdiv(P, 32)) splits = cdiv(P, 32);
  rpb = cdiv(P, splits);
  return cdiv(P, rpb);
}

// row split for bn_reduce_vec_kernel: <= 512 blocks (every block ends with 2C f64 atomics: 2048 blocks made the
// atomics, not the reads, the bottleneck), each at least 4 rows per thread group
static int split_rows_vec(int P, int cv, int& rpb) {
  const int groups = 256 / cv;
  int splits = 512;
  if (splits > cdiv(P, 4 * groups)) splits = cdiv(P, 4 * groups);
  if (splits < 1) splits = 1;
  rpb = cdiv(P, splits);
  return cdiv(P, rpb);
}

extern "C" {

// Training-mode forward.  sums: scratch double[mr_bn_scratch_doubles(C)] (MR_BN_COPIES accumulator copies of 2*C ...)
// (zeroed here unless flags bit 2 says the caller hands it over zeroed).  Saves mean/rstd (f32[C]) for backward.
long long mr_bn_scratch_doubles(int C) { return 2ll * C * MR_BN_COPIES + C; }   // + [2C] f32 of per-channel means (bwd)


// sums (f64 [MR_BN_COPIES][2][C], zeroed by the caller) += per-channel sum / sum of squares of x [P][C]: the statistics
// pass of mr_bn_fwd_train on its own (what mr_conv2d_fwd_stats falls back to); follow with mr_bn_fwd_train(flags bit 3).
int mr_bn_stats(int dtype, const void* x, double* sums, long long P, int C, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_bn_stats: C (%d) must be a multiple of %d", C, vec);
  MR_CHECK_ARG(P > 0 && P < (1ll << 31), "mr_bn_stats: bad P");
  int rpb;
  if (C / vec <= 256 && 256 % (C / vec) == 0) {
    const int splits = split_rows_vec((int)P, C / vec, rpb);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_reduce_vec_kernel<T, 0>), dim3(splits), dim3(256), 0, stream,
                                         (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                                         (const float*)nullptr, sums, 0, (int)P, C, rpb, MR_BN_COPIES));
  } else {
    const int splits = split_rows((int)P, C, rpb);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(cdiv(C, 64), splits), dim3(256), 0, stream,
                                         (const T*)x, sums, (int)P, C, (long long)C, rpb));
  }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_bn_fwd_train(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* save_mean, float* save_rstd, double* sums, const void* residual,
                    int relu, long long P, int C, float eps, float momentum, long long* num_batches_tracked,
                    hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_bn_fwd_train: C (%d) must be a multiple of %d", C, vec);
  MR_CHECK_ARG(P > 0 && P < (1ll << 31), "mr_bn_fwd_train: bad P");
  const int presum_zero = (relu >> 2) & 1;  // flags bit 2: the caller hands over an already zeroed `sums`
  const int have_stats = (relu >> 3) & 1;   // flags bit 3: `sums` already holds the statistics (mr_conv2d_fwd_stats / mr_bn_stats)
  relu &= 1;
  if (!presum_zero && !have_stats) (void)hipMemsetAsync(sums, 0, sizeof(double) * 2 * C * MR_BN_COPIES, stream);
  int rpb, ncopy = 1;
  if (have_stats) {
    ncopy = MR_BN_COPIES;
  } else if (C / vec <= 256 && 256 % (C / vec) == 0) {
    const int splits = split_rows_vec((int)P, C / vec, rpb);
    ncopy = MR_BN_COPIES;
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_reduce_vec_kernel<T, 0>), dim3(splits), dim3(256), 0, stream,
                                         (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                                         (const float*)nullptr, sums, 0, (int)P, C, rpb, ncopy));
  } else {
    const int splits = split_rows((int)P, C, rpb);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(cdiv(C, 64), splits), dim3(256), 0, stream,
                                         (const T*)x, sums, (int)P, C, (long long)C, rpb));
  }
  if (g_bn_fused && C % 64 == 0) {   // finalize folded into the apply pass (bn_apply_fused_kernel)
    const int rows = 256 / (64 / vec);
    const dim3 grid(bn_fused_grid_x(P, rows, C / 64), C / 64);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_fused_kernel<T>), grid, dim3(256), 0, stream, (const T*)x, (T*)y,
                                         (const double*)sums, ncopy, eps, momentum, save_mean, save_rstd, running_mean,
                                         running_var, num_batches_tracked, gamma, beta, (const T*)residual, relu,
                                         (int)P, C));
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, (const double*)sums, ncopy, (int)P, C,
                     eps, momentum, save_mean, save_rstd, running_mean, running_var, num_batches_tracked);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(P * (C / vec), 256)), dim3(256), 0,
                                       stream, (const T*)x, (T*)y, (const float*)save_mean, (const float*)save_rstd,
                                       gamma, beta, (const T*)residual, relu, P, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_bn_fwd_eval(int dtype, const void* x, void* y, const float* gamma, const float* beta,
                   const float* running_mean, const float* running_var, float* tmp_mean, float* tmp_rstd,
                   const void* residual, int relu, long long P, int C, float eps, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_bn_fwd_eval: C (%d) must be a multiple of %d", C, vec);
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, running_mean, running_var, eps,
                     tmp_mean, tmp_rstd, C);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(P * (C / vec), 256)), dim3(256), 0,
                                       stream, (const T*)x, (T*)y, (const float*)tmp_mean, (const float*)tmp_rstd,
                                       gamma, beta, (const T*)residual, relu, P, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// Backward of training-mode BN (+ optional fused ReLU / residual).  y is only read when the relu bit is set.
// flags: bit 0 = fused ReLU, bit 1 = accumulate (+=) into dgamma / dbeta instead of overwriting them, bit 2 = `sums` arrives
// zeroed, bit 3 = `sums` already holds the two reductions (the dgrad that produced dy accumulated them in its epilogue,
// mr_conv2d_dgrad_bnb): the reduction pass over dy / x / y is skipped.
int mr_bn_bwd(int dtype, const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
              const float* save_rstd, double* sums, void* dx, void* dres, float* dgamma, float* dbeta, int flags,
              long long P, int C, hipStream_t stream) {
  const int relu = flags & 1, accumulate = (flags >> 1) & 1, presum_zero = (flags >> 2) & 1;
  const int have_stats = (flags >> 3) & 1;   // `sums` already holds sum g' / sum g' xhat (mr_conv2d_dgrad_bnb's epilogue)
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_bn_bwd: C (%d) must be a multiple of %d", C, vec);
  MR_CHECK_ARG(P > 0 && P < (1ll << 31), "mr_bn_bwd: bad P");
  // one-pass path: the whole tensor in the registers of one resident grid (bn_bwd_onepass_kernel)
  if (!have_stats && g_bn_onepass && C % 64 == 0) {
    const int rows = 256 / (64 / vec);
    const int slabs = C / 64;
    const int cap = bn_onepass_cap();
    const long long g8 = (P + rows * 8 - 1) / (rows * 8), g2 = (P + rows * 2 - 1) / (rows * 2);
    const int rr = (g8 * slabs >= 256 || g2 * slabs > cap) ? 8 : 2;   // fewer rows per workgroup while that fills more CUs
    const long long gx = rr == 8 ? g8 : g2;
    if (cap > 0 && gx * slabs <= cap && P * C * (dtype == MR_F32 ? 4 : 2) < (1ll << 31)) {
      // arrival counters: one per slab, 512 bytes apart, behind the accumulator copies (the C doubles of the unfused path's
      // f32 [2C] area: unused here)
      unsigned* counters = (unsigned*)(sums + (size_t)2 * C * MR_BN_COPIES);
      if (!presum_zero) (void)hipMemsetAsync(sums, 0, sizeof(double) * (2 * C * MR_BN_COPIES + C), stream);
      const dim3 grid((unsigned)gx, slabs);
      if (rr == 8) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_onepass_kernel<T, 8>), grid, dim3(256), 0, stream, (const T*)dy,
                                             (const T*)x, (const T*)y, save_mean, save_rstd, gamma, sums, MR_BN_COPIES,
                                             counters, dgamma, dbeta, accumulate, (T*)dx, (T*)dres, relu, (int)P, C));
      } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_onepass_kernel<T, 2>), grid, dim3(256), 0, stream, (const T*)dy,
                                             (const T*)x, (const T*)y, save_mean, save_rstd, gamma, sums, MR_BN_COPIES,
                                             counters, dgamma, dbeta, accumulate, (T*)dx, (T*)dres, relu, (int)P, C));
      }
      MR_CHECK_LAUNCH();
      return MR_OK;
    }
  }
  if (!presum_zero && !have_stats) (void)hipMemsetAsync(sums, 0, sizeof(double) * 2 * C * MR_BN_COPIES, stream);
  int rpb, ncopy = 1;
  if (have_stats) {
    ncopy = MR_BN_COPIES;
  } else if (C / vec <= 256 && 256 % (C / vec) == 0) {
    const int splits = split_rows_vec((int)P, C / vec, rpb);
    ncopy = MR_BN_COPIES;
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_reduce_vec_kernel<T, 1>), dim3(splits), dim3(256), 0, stream,
                                         (const T*)dy, (const T*)x, (const T*)y, save_mean, save_rstd, sums, relu,
                                         (int)P, C, rpb, ncopy));
  } else {
    const int splits = split_rows((int)P, C, rpb);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(cdiv(C, 64), splits), dim3(256), 0, stream,
                                         (const T*)dy, (const T*)x, (const T*)y, save_mean, save_rstd, sums, relu,
                                         (int)P, C, rpb));
  }
  if (g_bn_fused && C % 64 == 0) {   // finalize folded into the apply pass (bn_bwd_apply_fused_kernel)
    const int rows = 256 / (64 / vec);
    const dim3 grid(bn_fused_grid_x(P, rows, C / 64), C / 64);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_fused_kernel<T>), grid, dim3(256), 0, stream, (const T*)dy,
                                         (const T*)x, (const T*)y, save_mean, save_rstd, gamma, (const double*)sums, ncopy,
                                         dgamma, dbeta, accumulate, (T*)dx, (T*)dres, relu, (int)P, C));
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  float* sbg = (float*)(sums + (size_t)2 * C * MR_BN_COPIES);   // [2C] f32 behind the accumulator copies
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, (const double*)sums, ncopy, P, C,
                     sbg, dgamma, dbeta, accumulate);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(grid_for(P * (C / vec), 256)), dim3(256), 0,
                                       stream, (const T*)dy, (const T*)x, (const T*)y, save_mean, save_rstd, gamma,
                                       (const float*)sbg, (T*)dx, (T*)dres, relu, P, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_maxpool_fwd(int dtype, const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int kh,
                   int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_maxpool_fwd: C (%d) must be a multiple of %d", C, vec);
  MR_CHECK_ARG(kh * kw <= 255, "mr_maxpool_fwd: window too large");
  MR_CHECK_ARG(Ho == (H + 2 * ph - kh) / sh + 1 && Wo == (W + 2 * pw - kw) / sw + 1,
               "mr_maxpool_fwd: output size inconsistent");
  const long long total = (long long)N * Ho * Wo * (C / vec);
  if ((long long)N * Ho <= 65535 && MR_TUNE(pool_fixed)) {   // one (n, ho) row per blockIdx.y
    const int cv = C / vec;
    const dim3 grid(cdiv(Wo * cv, 256), N * Ho);
#define MR_POOL_FIXED(KH_, KW_, SH_, SW_)                                                                            \
  if (kh == KH_ && kw == KW_ && sh == SH_ && sw == SW_) {                                                            \
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_fwd_fixed_kernel<T, KH_, KW_, SH_, SW_>), grid, dim3(256), 0,       \
                                         stream, (const T*)x, (T*)y, idx, H, W, cv, ph, pw, Ho, Wo));                \
    MR_CHECK_LAUNCH();                                                                                               \
    return MR_OK;                                                                                                    \
  }
    MR_POOL_FIXED(2, 2, 2, 2) MR_POOL_FIXED(2, 2, 2, 1) MR_POOL_FIXED(3, 3, 2, 2)
#undef MR_POOL_FIXED
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)x, (T*)y, idx, N, H, W, C, kh, kw, sh, sw, ph, pw, Ho, Wo));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_maxpool_bwd(int dtype, const void* dy, const unsigned char* idx, const void* relu_y, void* dx, int N, int H,
                   int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo,
                   hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_maxpool_bwd: C (%d) must be a multiple of %d", C, vec);
  const long long total = (long long)N * H * W * (C / vec);
  const int cv = C / vec;
  if (kh == 2 && kw == 2 && sh == 2 && sw == 2 && ph == 0 && pw == 0 && H == 2 * Ho && W == 2 * Wo &&
      (long long)N * Ho <= 65535 && MR_TUNE(pool_fixed)) {   // organised by pooled element (maxpool_bwd_2x2s2_kernel)
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_2x2s2_kernel<T>), dim3(cdiv(Wo * cv, 256), N * Ho), dim3(256), 0, stream,
                                         (const T*)dy, idx, (const T*)relu_y, (T*)dx, H, W, cv, Ho, Wo));
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  if (kh == 2 && kw == 2 && sh == 2 && sw == 1 && ph == 0 && (pw == 0 || pw == 1) && H == 2 * Ho && Wo == W + 2 * pw - 1 &&
      (long long)N * Ho <= 65535 && MR_TUNE(pool_fixed)) {   // both input rows of a window row per thread
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_2x2s21_kernel<T>), dim3(cdiv(W * cv, 256), N * Ho), dim3(256), 0, stream,
                                         (const T*)dy, idx, (const T*)relu_y, (T*)dx, H, W, cv, pw, Ho, Wo));
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  if ((long long)N * H <= 65535) {   // one (n, h) row per blockIdx.y
    const dim3 grid(cdiv(W * cv, 256), N * H);
#define MR_POOL_FIXED(KH_, KW_, SH_, SW_)                                                                            \
  if (kh == KH_ && kw == KW_ && sh == SH_ && sw == SW_) {                                                            \
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_fixed_kernel<T, KH_, KW_, SH_, SW_>), grid, dim3(256), 0,       \
                                         stream, (const T*)dy, idx, (const T*)relu_y, (T*)dx, H, W, cv, ph, pw, Ho,  \
                                         Wo));                                                                       \
    MR_CHECK_LAUNCH();                                                                                               \
    return MR_OK;                                                                                                    \
  }
    MR_POOL_FIXED(2, 2, 2, 2) MR_POOL_FIXED(2, 2, 2, 1) MR_POOL_FIXED(3, 3, 2, 2)
#undef MR_POOL_FIXED
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)dy, idx, (const T*)relu_y, (T*)dx, N, H, W, C, kh, kw, sh, sw, ph,
                                       pw, Ho, Wo));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
