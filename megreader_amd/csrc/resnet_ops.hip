// HBM-bound helpers for the ResNet50-dilated / PPM backbone and the 2D-CTC head (NHWC tensors):
//   adaptive average pooling  (backbones/ppm.py:13, nn.AdaptiveAvgPool2d(scale))
//   bilinear resize, align_corners=False  (backbones/ppm.py:37-40, backbones/fpn_top_down.py:19)
//   channel-slice copy (torch.cat / its backward, backbones/ppm.py:41)
//   per-(sample, channel) scaling (nn.Dropout2d, backbones/ppm.py:27)
//   2D-CTC head: log(max(softmax_H(mask) * softmax_C(classify), tiny)) -> [W,H,N,C]  (decoders/ctc_decoder2d.py:16-45)
#include "common.h"
#include "../../include/megreader_hip.h"

#define MR_POOL_MULTI_MAX 8   /* scales per mr_adaptive_avgpool_multi_* call */
#define MR_POOL_MULTI_BINS 256 /* sum of oh*ow over the scales */

namespace mr {

static inline int grid_for(long long n, int block, int max_blocks = 16384) {
  long long b = (n + block - 1) / block;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ int bin_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int bin_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

// ---------------------------------------------------------------- adaptive average pool
template <typename T>
__global__ void adaptive_avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                            int OH, int OW) {
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long q = i / C;
    const int ow = (int)(q % OW);
    q /= OW;
    const int oh = (int)(q % OH);
    const int n = (int)(q / OH);
    const int h0 = bin_start(oh, H, OH), h1 = bin_end(oh, H, OH);
    const int w0 = bin_start(ow, W, OW), w1 = bin_end(ow, W, OW);
    float s = 0.f;
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) s += to_f32(x[(((long long)n * H + h) * W + w) * C + c]);
    y[i] = from_f32<T>(s / (float)((h1 - h0) * (w1 - w0)));
  }
}

// One thread = one 16-byte channel vector of one input pixel.  Bin oh covers rows [floor(oh*H/OH), ceil((oh+1)*H/OH)),
// so the bins containing row h form a short contiguous range computed in closed form (the first version walked all
// OH x OW bins with two integer divisions each, per 2-byte element: 320 us on the PPM's [256,4,16,2048] map
// instead of ~20).
template <typename T>
__global__ void adaptive_avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C,
                                            int OH, int OW) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = (long long)N * H * W * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    long long q = i / cv;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H);
    const int n = (int)(q / H);
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    // row h lies in bin oh  <=>  floor(h*OH/H) <= oh <= ceil((h+1)*OH/H) - 1   (exact, also when OH > H)
    const int oh_lo = h * OH / H, oh_hi = min(OH - 1, ((h + 1) * OH + H - 1) / H - 1);
    const int ow_lo = w * OW / W, ow_hi = min(OW - 1, ((w + 1) * OW + W - 1) / W - 1);
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int h0 = bin_start(oh, H, OH), h1 = bin_end(oh, H, OH);
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int w0 = bin_start(ow, W, OW), w1 = bin_end(ow, W, OW);
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
        const uint4 g = ((const uint4*)dy)[(((long long)n * OH + oh) * OW + ow) * cv + c];
        const T* pg = (const T*)&g;
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] += to_f32(pg[j]) * inv;
      }
    }
    uint4 o;
    T* po = (T*)&o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(s[j]);
    ((uint4*)dx)[i] = o;
  }
}

// ---------------------------------------------------------------- pyramid pooling: several adaptive average pools of one map
// The PPM (reference backbones/ppm.py:13-20,36-40) pools the same [N, H, W, C] map to 1x1, 2x2, 3x3 and 6x6: four forward
// passes over the map and, backward, four full-size input gradients that autograd then adds up (the map is the largest
// activation of the head: 67 MB at N = 256).  Forward: one workgroup stages the H*W pixels of a 128-byte channel slice in LDS and
// computes every bin of every scale from it (same summation order as adaptive_avgpool_fwd_kernel: bit-identical outputs).
// Backward: one pass writes dx = sum over scales, accumulated in f32 and rounded once.
struct PoolMultiArgs {
  int nsc;
  int oh[MR_POOL_MULTI_MAX], ow[MR_POOL_MULTI_MAX];
  void* y[MR_POOL_MULTI_MAX];       // forward outputs / backward incoming gradients, [N][oh][ow][C]
};

// Bin tables are built once per workgroup in LDS: computing bin_start / bin_end (integer divisions) per bin per thread made
// both kernels VALU-bound (113 + 139 us on the [256,4,16,2048] map, r05 trace) instead of one pass over the map.
__device__ __forceinline__ int pool_multi_find(const PoolMultiArgs& a, int b, int& local) {
  int s = 0;
  while (s < a.nsc - 1 && b >= a.oh[s] * a.ow[s]) { b -= a.oh[s] * a.ow[s]; ++s; }
  local = b;
  return s;
}

template <typename T>
__global__ __launch_bounds__(256) void adaptive_avgpool_multi_fwd_kernel(const T* __restrict__ x, PoolMultiArgs a, int N,
                                                                         int H, int W, int C, int nbins) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int CPB = 128 / (int)sizeof(T);          // channels per workgroup: one 128-byte line per pixel
  extern __shared__ uint4 pool_tile[];               // [H*W][8] vectors, then ushort4 bins[nbins] = (h0, h1, w0, w1)
  const int n = blockIdx.y, c0 = blockIdx.x * CPB;
  const int HW = H * W;
  ushort4* bins = (ushort4*)(pool_tile + HW * 8);
  for (int b = threadIdx.x; b < nbins; b += 256) {
    int l;
    const int s = pool_multi_find(a, b, l);
    const int OH = a.oh[s], OW = a.ow[s];
    const int oh = l / OW, ow = l - oh * OW;
    bins[b] = make_ushort4((unsigned short)bin_start(oh, H, OH), (unsigned short)bin_end(oh, H, OH),
                           (unsigned short)bin_start(ow, W, OW), (unsigned short)bin_end(ow, W, OW));
  }
  for (int i = threadIdx.x; i < HW * 8; i += 256) {
    const int pix = i >> 3, v = i & 7;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (c0 + v * VEC < C) val = *(const uint4*)(x + ((long long)n * HW + pix) * C + c0 + v * VEC);
    pool_tile[i] = val;
  }
  __syncthreads();
  const T* tile = (const T*)pool_tile;
  const int c = threadIdx.x % CPB, sub = threadIdx.x / CPB, nsub = 256 / CPB;
  if (c0 + c >= C) return;
  int base = 0;
  for (int s = 0; s < a.nsc; ++s) {
    const int nb = a.oh[s] * a.ow[s];
    T* __restrict__ y = (T*)a.y[s];
    for (int b = sub; b < nb; b += nsub) {
      const ushort4 e = bins[base + b];
      float acc = 0.f;
      for (int h = e.x; h < e.y; ++h)
        for (int w = e.z; w < e.w; ++w) acc += to_f32(tile[(h * W + w) * CPB + c]);
      y[((long long)n * nb + b) * C + c0 + c] = from_f32<T>(acc / (float)((e.y - e.x) * (e.w - e.z)));
    }
    base += nb;
  }
}

// One workgroup = one image x one 128-byte channel slice: the slice of every bin's gradient is staged in LDS (nbins x 128 B)
// next to the row -> bins and column -> bins ranges; each thread then sums the bins covering its pixel in the order
// (scale, oh, ow) of the element-wise kernel above and writes one 16-byte vector.
template <typename T>
__global__ __launch_bounds__(256) void adaptive_avgpool_multi_bwd_kernel(PoolMultiArgs a, T* __restrict__ dx, int N, int H,
                                                                         int W, int C, int nbins) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int CPB = 128 / (int)sizeof(T);
  extern __shared__ uint4 pool_tile[];               // [nbins][8] vectors | float inv[nbins] | ushort2 rows[nsc*H] | cols[nsc*W]
  const int n = blockIdx.y, c0 = blockIdx.x * CPB;
  const int HW = H * W;
  float* inv = (float*)(pool_tile + nbins * 8);
  ushort2* rows = (ushort2*)(inv + nbins);
  ushort2* cols = rows + a.nsc * H;
  for (int b = threadIdx.x; b < nbins; b += 256) {
    int l;
    const int s = pool_multi_find(a, b, l);
    const int OH = a.oh[s], OW = a.ow[s];
    const int oh = l / OW, ow = l - oh * OW;
    inv[b] = 1.f / (float)((bin_end(oh, H, OH) - bin_start(oh, H, OH)) * (bin_end(ow, W, OW) - bin_start(ow, W, OW)));
  }
  for (int i = threadIdx.x; i < a.nsc * (H + W); i += 256) {
    // row h lies in bin oh  <=>  floor(h*OH/H) <= oh <= ceil((h+1)*OH/H) - 1   (see adaptive_avgpool_bwd_kernel)
    if (i < a.nsc * H) {
      const int s = i / H, h = i - s * H, OH = a.oh[s];
      rows[i] = make_ushort2((unsigned short)(h * OH / H), (unsigned short)min(OH - 1, ((h + 1) * OH + H - 1) / H - 1));
    } else {
      const int j = i - a.nsc * H;
      const int s = j / W, w = j - s * W, OW = a.ow[s];
      cols[j] = make_ushort2((unsigned short)(w * OW / W), (unsigned short)min(OW - 1, ((w + 1) * OW + W - 1) / W - 1));
    }
  }
  {
    int base = 0;
    for (int s = 0; s < a.nsc; ++s) {
      const int nb = a.oh[s] * a.ow[s];
      const T* __restrict__ dy = (const T*)a.y[s];
      for (int i = threadIdx.x; i < nb * 8; i += 256) {
        const int b = i >> 3, v = i & 7;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (c0 + v * VEC < C) val = *(const uint4*)(dy + ((long long)n * nb + b) * C + c0 + v * VEC);
        pool_tile[(base + b) * 8 + v] = val;
      }
      base += nb;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HW * 8; i += 256) {
    const int pix = i >> 3, v = i & 7;
    if (c0 + v * VEC >= C) continue;
    const int h = pix / W, w = pix - h * W;
    float sum[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) sum[j] = 0.f;
    int base = 0;
    for (int s = 0; s < a.nsc; ++s) {
      const int OW = a.ow[s];
      const ushort2 rr = rows[s * H + h], cc = cols[s * W + w];
      for (int oh = rr.x; oh <= rr.y; ++oh)
        for (int ow = cc.x; ow <= cc.y; ++ow) {
          const int b = base + oh * OW + ow;
          const float iv = inv[b];
          const uint4 g = pool_tile[b * 8 + v];
          const T* pg = (const T*)&g;
#pragma unroll
          for (int j = 0; j < VEC; ++j) sum[j] += to_f32(pg[j]) * iv;
        }
      base += a.oh[s] * OW;
    }
    uint4 o;
    T* po = (T*)&o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(sum[j]);
    *(uint4*)(dx + ((long long)n * HW + pix) * C + c0 + v * VEC) = o;
  }
}

// ---------------------------------------------------------------- bilinear resize (align_corners = False)
__device__ __forceinline__ void bilin_src(int o, int in, int out, int& i0, int& i1, float& l1) {
  // PyTorch area_pixel_compute_source_index: src = (o + 0.5) * in/out - 0.5, clamped at 0
  float src = ((float)o + 0.5f) * ((float)in / (float)out) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

// y[n, oh, ow, coff + c] (ld = ldy) (+)= bilinear(x)[...]; accumulate != 0 adds into y (FPN top-down add)
template <typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int OH,
                                    int OW, int ldy, int coff, int accumulate) {
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long q = i / C;
    const int ow = (int)(q % OW);
    q /= OW;
    const int oh = (int)(q % OH);
    const int n = (int)(q / OH);
    int h0, h1, w0, w1;
    float lh, lw;
    bilin_src(oh, H, OH, h0, h1, lh);
    bilin_src(ow, W, OW, w0, w1, lw);
    const T* xb = x + (long long)n * H * W * C + c;
    const float v = (1.f - lh) * ((1.f - lw) * to_f32(xb[((long long)h0 * W + w0) * C]) +
                                  lw * to_f32(xb[((long long)h0 * W + w1) * C])) +
                    lh * ((1.f - lw) * to_f32(xb[((long long)h1 * W + w0) * C]) +
                          lw * to_f32(xb[((long long)h1 * W + w1) * C]));
    T* dst = y + (((long long)n * OH + oh) * OW + ow) * ldy + coff + c;
    *dst = from_f32<T>(accumulate ? to_f32(*dst) + v : v);
  }
}

// dx[n,h,w,c] = sum over output pixels of their interpolation weight on (h,w) times dy (gather form, no atomics)
template <typename T>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int OH,
                                    int OW, int lddy, int coff) {
  const long long total = (long long)N * H * W * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long q = i / C;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H);
    const int n = (int)(q / H);
    float s = 0.f;
    for (int oh = 0; oh < OH; ++oh) {
      int h0, h1;
      float lh;
      bilin_src(oh, H, OH, h0, h1, lh);
      float wh = 0.f;
      if (h0 == h) wh += 1.f - lh;
      if (h1 == h) wh += lh;
      if (wh == 0.f) continue;
      for (int ow = 0; ow < OW; ++ow) {
        int w0, w1;
        float lw;
        bilin_src(ow, W, OW, w0, w1, lw);
        float ww = 0.f;
        if (w0 == w) ww += 1.f - lw;
        if (w1 == w) ww += lw;
        if (ww == 0.f) continue;
        s += wh * ww * to_f32(dy[(((long long)n * OH + oh) * OW + ow) * lddy + coff + c]);
      }
    }
    dx[i] = from_f32<T>(s);
  }
}

// 16-byte-vector forms (C, ldy, coff multiples of one vector): same arithmetic per element as the scalar kernels above.
template <typename T>
__global__ void bilinear_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int OH,
                                        int OW, int ldy, int coff, int accumulate) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = (long long)N * OH * OW * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    long long q = i / cv;
    const int ow = (int)(q % OW);
    q /= OW;
    const int oh = (int)(q % OH);
    const int n = (int)(q / OH);
    int h0, h1, w0, w1;
    float lh, lw;
    bilin_src(oh, H, OH, h0, h1, lh);
    bilin_src(ow, W, OW, w0, w1, lw);
    const T* xb = x + (long long)n * H * W * C + c;
    const uint4 v00 = *(const uint4*)(xb + ((long long)h0 * W + w0) * C), v01 = *(const uint4*)(xb + ((long long)h0 * W + w1) * C);
    const uint4 v10 = *(const uint4*)(xb + ((long long)h1 * W + w0) * C), v11 = *(const uint4*)(xb + ((long long)h1 * W + w1) * C);
    const T *p00 = (const T*)&v00, *p01 = (const T*)&v01, *p10 = (const T*)&v10, *p11 = (const T*)&v11;
    T* dst = y + (((long long)n * OH + oh) * OW + ow) * ldy + coff + c;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (accumulate) o = *(const uint4*)dst;
    T* po = (T*)&o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float v = (1.f - lh) * ((1.f - lw) * to_f32(p00[j]) + lw * to_f32(p01[j])) +
                      lh * ((1.f - lw) * to_f32(p10[j]) + lw * to_f32(p11[j]));
      po[j] = from_f32<T>(accumulate ? to_f32(po[j]) + v : v);
    }
    *(uint4*)dst = o;
  }
}

// Output pixels touching input index i form a contiguous range (the source coordinate is monotone in o): a conservative
// float estimate of it, widened by one either side; the exact index test of the scalar kernel is kept inside.
__device__ __forceinline__ void bilin_touch_range(int i, int in, int out, int& lo, int& hi) {
  const float inv = (float)out / (float)in;
  lo = max(0, (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1);
  hi = min(out - 1, (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1);
}

template <typename T>
__global__ void bilinear_bwd_vec_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int OH,
                                        int OW, int lddy, int coff) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = (long long)N * H * W * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    long long q = i / cv;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H);
    const int n = (int)(q / H);
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    int oh_lo, oh_hi, ow_lo, ow_hi;
    bilin_touch_range(h, H, OH, oh_lo, oh_hi);
    bilin_touch_range(w, W, OW, ow_lo, ow_hi);
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      int h0, h1;
      float lh;
      bilin_src(oh, H, OH, h0, h1, lh);
      float wh = 0.f;
      if (h0 == h) wh += 1.f - lh;
      if (h1 == h) wh += lh;
      if (wh == 0.f) continue;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        int w0, w1;
        float lw;
        bilin_src(ow, W, OW, w0, w1, lw);
        float ww = 0.f;
        if (w0 == w) ww += 1.f - lw;
        if (w1 == w) ww += lw;
        if (ww == 0.f) continue;
        const uint4 g = *(const uint4*)(dy + (((long long)n * OH + oh) * OW + ow) * lddy + coff + c);
        const T* pg = (const T*)&g;
        const float wt = wh * ww;
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] += wt * to_f32(pg[j]);
      }
    }
    uint4 o;
    T* po = (T*)&o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(s[j]);
    *(uint4*)(dx + ((((long long)n * H + h) * W + w) * C + c)) = o;
  }
}

// ---------------------------------------------------------------- channel-slice copy: dst[p, doff + c] = src[p, soff + c]
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int lds, int soff, T* __restrict__ dst, int ldd,
                                     int doff, long long P, int C) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = P * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    const long long p = i / cv;
    *(uint4*)(dst + p * ldd + doff + c) = *(const uint4*)(src + p * lds + soff + c);
  }
}

// ---------------------------------------------------------------- gradient of a spatial sub-sampling x[:, ::sh, ::sw, :]
// dx[n, h, w, :] = (h % sh == 0 && w % sw == 0 && h / sh < Ho && w / sw < Wo) ? dxs[n, h / sh, w / sw, :] : 0 -- one pass over
// dx.  The data gradient of a strided 1x1 convolution (the downsample branch of a ResNet stage, backbones/resnet.py:204-213) is
// the dense dgrad on the sub-sampled grid followed by this: the implicit-GEMM dgrad spent its MFMAs on the 3/4 of the output
// rows that are zero (2048 x 1024 x 2048 at stride 2: 52 us = 41 TFLOP/s).
template <typename T>
__global__ void scatter_strided_kernel(const T* __restrict__ dxs, T* __restrict__ dx, int N, int H, int W, int C, int sh,
                                       int sw, int Ho, int Wo) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = (long long)N * H * W * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    long long p = i / cv;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const long long n = p / H;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (h % sh == 0 && w % sw == 0 && h / sh < Ho && w / sw < Wo)
      v = ((const uint4*)dxs)[((n * Ho + h / sh) * Wo + w / sw) * cv + c];
    ((uint4*)dx)[i] = v;
  }
}

// ---------------------------------------------------------------- y[n,p,c] = x[n,p,c] * scale[n,c]  (Dropout2d)
template <typename T>
__global__ void scale_channels_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y,
                                      int N, long long HW, int C) {
  const long long total = (long long)N * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int n = (int)(i / (HW * C));
    y[i] = from_f32<T>(to_f32(x[i]) * scale[(long long)n * C + c]);
  }
}

// ---------------------------------------------------------------- 2D-CTC head
// mask logits a[n,h,w] (ld = lda, channel 0), class logits z[n,h,w,c] (ld = ldz).  One wave per (n, w) column:
//   m[h] = softmax_h(a), p[h,c] = softmax_c(z[h]);  lp[w,h,n,c] = log(max(m[h]*p[h,c], tiny))   (f32 out)
// also stores m [N,H,W] and p [N,H,W,C] (f32) for the backward / eval outputs.
template <typename T>
__global__ void ctc2d_head_fwd_kernel(const T* __restrict__ a, int lda, const T* __restrict__ z, int ldz,
                                      float* __restrict__ lp, float* __restrict__ m_out, float* __restrict__ p_out,
                                      int N, int H, int W, int C, float tiny) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // col = n*W + w
  if (col >= N * W) return;
  const int n = col / W, w = col - n * W;
  // softmax over H of the mask logits (H is small: lanes stride over it)
  float mx = -INFINITY;
  for (int h = lane; h < H; h += 64) mx = fmaxf(mx, to_f32(a[(((long long)n * H + h) * W + w) * lda]));
  mx = wave_max(mx);
  float se = 0.f;
  for (int h = lane; h < H; h += 64) se += expf(to_f32(a[(((long long)n * H + h) * W + w) * lda]) - mx);
  se = wave_sum(se);
  for (int h = 0; h < H; ++h) {
    const long long pix = ((long long)n * H + h) * W + w;
    const float mh = expf(to_f32(a[pix * lda]) - mx) / se;
    if (lane == 0) m_out[pix] = mh;
    const T* zr = z + pix * ldz;
    float cm = -INFINITY;
    for (int c = lane; c < C; c += 64) cm = fmaxf(cm, to_f32(zr[c]));
    cm = wave_max(cm);
    float cs = 0.f;
    for (int c = lane; c < C; c += 64) cs += expf(to_f32(zr[c]) - cm);
    cs = wave_sum(cs);
    float* lrow = lp + (((long long)w * H + h) * N + n) * C;
    for (int c = lane; c < C; c += 64) {
      const float pc = expf(to_f32(zr[c]) - cm) / cs;
      p_out[pix * C + c] = pc;
      lrow[c] = logf(fmaxf(mh * pc, tiny));
    }
  }
}

// backward of the head: g = d loss / d lp [W,H,N,C].  With k = [m*p > tiny] (the clamp passes no gradient):
//   da[h] = G_h - m[h] * sum_h' G_h',  G_h = sum_c g*k ;   dz[h,c] = g*k - p[h,c] * G_h
template <typename T>
__global__ void ctc2d_head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ m,
                                      const float* __restrict__ p, T* __restrict__ da, int ldda, T* __restrict__ dz,
                                      int lddz, int N, int H, int W, int C, float tiny) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (col >= N * W) return;
  const int n = col / W, w = col - n * W;
  float Gtot = 0.f;
  for (int h = 0; h < H; ++h) {
    const long long pix = ((long long)n * H + h) * W + w;
    const float mh = m[pix];
    const float* grow = g + (((long long)w * H + h) * N + n) * C;
    float Gh = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float pc = p[pix * C + c];
      if (mh * pc > tiny) Gh += grow[c];
    }
    Gh = wave_sum(Gh);
    Gtot += Gh;
    T* dzr = dz + pix * lddz;
    for (int c = lane; c < C; c += 64) {
      const float pc = p[pix * C + c];
      const float gk = (mh * pc > tiny) ? grow[c] : 0.f;
      dzr[c] = from_f32<T>(gk - pc * Gh);
    }
  }
  // second pass for da needs Gtot: recompute G_h (cheap)
  for (int h = 0; h < H; ++h) {
    const long long pix = ((long long)n * H + h) * W + w;
    const float mh = m[pix];
    const float* grow = g + (((long long)w * H + h) * N + n) * C;
    float Gh = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float pc = p[pix * C + c];
      if (mh * pc > tiny) Gh += grow[c];
    }
    Gh = wave_sum(Gh);
    if (lane == 0) da[pix * ldda] = from_f32<T>(Gh - mh * Gtot);
  }
}

// Nearest-neighbour upsampling by an integer factor (nn.Upsample(scale_factor=s, mode='nearest') of the DB head,
// decoders/seg_detector.py:22-43): y[n, oh, ow, coff + c] = x[n, oh / s, ow / s, c] (+ add[n, oh, ow, c]); 16-byte vectors.
template <typename T>
__global__ void nearest_up_fwd_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ y, int N, int H,
                                      int W, int C, int s, int ldy, int coff) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC, OH = H * s, OW = W * s;
  const long long total = (long long)N * OH * OW * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    long long q = i / cv;
    const int ow = (int)(q % OW);
    q /= OW;
    const int oh = (int)(q % OH);
    const int n = (int)(q / OH);
    uint4 v = *(const uint4*)(x + (((long long)n * H + oh / s) * W + ow / s) * C + c);
    if (add) {
      const uint4 a = *(const uint4*)(add + (((long long)n * OH + oh) * OW + ow) * C + c);
      const T* pa = (const T*)&a;
      T* pv = (T*)&v;
#pragma unroll
      for (int j = 0; j < VEC; ++j) pv[j] = from_f32<T>(to_f32(pv[j]) + to_f32(pa[j]));
    }
    *(uint4*)(y + (((long long)n * OH + oh) * OW + ow) * ldy + coff + c) = v;
  }
}

// ---- ConvTranspose2d(k = 2, s = 2) as a GEMM + depth-to-space (decoders/seg_detector.py:66-79: the DB heads' two deconvolutions).
// The GEMM leaves y2[p, co*4 + i*2 + j] (columns in the WEIGHT's own order [Cin][Cout][2][2], so the weight gradient of the
// transposed GEMM lands in the parameter's layout); this pass writes y[n, 2h+i, 2w+j, co] = y2[p, co, i, j] + bias[co].  A thread
// owns one pixel p and a run of VEC output channels: 4*VEC contiguous source elements, four VEC-wide stores.
template <typename T>
__global__ void deconv2x2_d2s_kernel(const T* __restrict__ y2, int ld2, const float* __restrict__ bias, T* __restrict__ y, int N,
                                     int H, int W, int C) {
  constexpr int VEC = VecOf<T>::N;
  const int cg = (C + VEC - 1) / VEC;
  const long long total = (long long)N * H * W * cg;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(t % cg);
    const long long p = t / cg;
    const int w = (int)(p % W);
    const long long r = p / W;
    const int h = (int)(r % H), n = (int)(r / H);
    const int c0 = g * VEC, nc = min(VEC, C - c0);
    const T* src = y2 + p * ld2 + (long long)c0 * 4;
    if (nc == VEC && (C % VEC) == 0 && (ld2 % VEC) == 0) {   // whole vectors: 4 x 16-byte loads, 4 x 16-byte stores
      uint4 in[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) in[q] = ((const uint4*)src)[q];
      const T* pin = (const T*)in;
      float bv[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) bv[j] = bias ? bias[c0 + j] : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint4 o;
        T* po = (T*)&o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(to_f32(pin[j * 4 + k]) + bv[j]);
        *(uint4*)(y + ((((long long)n * 2 * H + 2 * h + (k >> 1)) * 2 * W) + 2 * w + (k & 1)) * C + c0) = o;
      }
      continue;
    }
    for (int k = 0; k < 4; ++k) {
      T* dst = y + ((((long long)n * 2 * H + 2 * h + (k >> 1)) * 2 * W) + 2 * w + (k & 1)) * C + c0;
      for (int j = 0; j < nc; ++j) dst[j] = from_f32<T>(to_f32(src[j * 4 + k]) + (bias ? bias[c0 + j] : 0.f));
    }
  }
}

// the inverse gather for the backward pass: dy2[p, co*4 + i*2 + j] = dy[n, 2h+i, 2w+j, co]; columns 4C .. ld2-1 (padding of the
// GEMM operand to whole vectors) are written as zeros
template <typename T>
__global__ void deconv2x2_s2d_kernel(const T* __restrict__ dy, T* __restrict__ dy2, int ld2, int N, int H, int W, int C) {
  constexpr int VEC = VecOf<T>::N;
  const int cg = (C + VEC - 1) / VEC;
  const long long total = (long long)N * H * W * cg;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(t % cg);
    const long long p = t / cg;
    const int w = (int)(p % W);
    const long long r = p / W;
    const int h = (int)(r % H), n = (int)(r / H);
    const int c0 = g * VEC, nc = min(VEC, C - c0);
    T* dst = dy2 + p * ld2 + (long long)c0 * 4;
    if (nc == VEC && (C % VEC) == 0 && (ld2 % VEC) == 0) {
      uint4 in[4], out[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        in[k] = *(const uint4*)(dy + ((((long long)n * 2 * H + 2 * h + (k >> 1)) * 2 * W) + 2 * w + (k & 1)) * C + c0);
      const T* pin = (const T*)in;
      T* pout = (T*)out;
#pragma unroll
      for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) pout[j * 4 + k] = pin[k * VEC + j];
#pragma unroll
      for (int q = 0; q < 4; ++q) ((uint4*)dst)[q] = out[q];
    } else {
      for (int k = 0; k < 4; ++k) {
        const T* src = dy + ((((long long)n * 2 * H + 2 * h + (k >> 1)) * 2 * W) + 2 * w + (k & 1)) * C + c0;
        for (int j = 0; j < nc; ++j) dst[j * 4 + k] = src[j];
      }
    }
    if (g == cg - 1)
      for (int c = 4 * C; c < ld2; ++c) dy2[p * ld2 + c] = from_f32<T>(0.f);
  }
}

// dx[n, h, w, c] = sum over the s x s block of dy[n, h*s + i, w*s + j, coff + c]   (gather form, no atomics)
template <typename T>
__global__ void nearest_up_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int s,
                                      int lddy, int coff) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC, OW = W * s, OH = H * s;
  const long long total = (long long)N * H * W * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    long long q = i / cv;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H);
    const int n = (int)(q / H);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int a = 0; a < s; ++a)
      for (int b = 0; b < s; ++b) {
        const uint4 v = *(const uint4*)(dy + (((long long)n * OH + h * s + a) * OW + w * s + b) * lddy + coff + c);
        const T* pv = (const T*)&v;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += to_f32(pv[j]);
      }
    uint4 o;
    T* po = (T*)&o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(acc[j]);
    *(uint4*)(dx + (((long long)n * H + h) * W + w) * C + c) = o;
  }
}

}  // namespace mr

using namespace mr;

#define DISPATCH_T(dtype, ...)                                   \
  if ((dtype) == MR_F32) { typedef float T; __VA_ARGS__; }       \
  else if ((dtype) == MR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { mr::set_error("bad dtype %d", (dtype)); return MR_ERR_DTYPE; }

extern "C" {

int mr_adaptive_avgpool_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, int OH, int OW,
                            hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "mr_adaptive_avgpool_fwd: bad shape");
  const long long total = (long long)N * OH * OW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_avgpool_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                       stream, (const T*)x, (T*)y, N, H, W, C, OH, OW));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_adaptive_avgpool_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int OH, int OW,
                            hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_adaptive_avgpool_bwd: C (%d) must be a multiple of %d", C, vec);
  const long long total = (long long)N * H * W * (C / vec);
  DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_avgpool_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                       stream, (const T*)dy, (T*)dx, N, H, W, C, OH, OW));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// Several adaptive average pools of ONE map in one launch each way (pyramid pooling, reference backbones/ppm.py:13-20,36-40).
// ys / dys: host array of nscales device pointers [N][oh[i]][ow[i]][C]; forward outputs equal mr_adaptive_avgpool_fwd's bit for
// bit; backward writes dx = sum_i adaptive_avgpool_bwd(dys[i]) accumulated in f32.  C % (16 / sizeof T) == 0, H*W*128 B <= 64 KB.
int mr_adaptive_avgpool_multi_fwd(int dtype, const void* x, void* const* ys, const int* oh, const int* ow, int nscales, int N,
                                  int H, int W, int C, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(nscales > 0 && nscales <= MR_POOL_MULTI_MAX && N > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0,
               "mr_adaptive_avgpool_multi_fwd: bad arguments");
  MR_CHECK_ARG((long long)H * W * 128 <= 61440 && H < 65536 && W < 65536,
               "mr_adaptive_avgpool_multi_fwd: H*W (%d) too large for the LDS tile", H * W);
  PoolMultiArgs a;
  a.nsc = nscales;
  for (int i = 0; i < nscales; ++i) {
    MR_CHECK_ARG(ys[i] != nullptr && oh[i] > 0 && ow[i] > 0, "mr_adaptive_avgpool_multi_fwd: bad scale %d", i);
    a.oh[i] = oh[i]; a.ow[i] = ow[i]; a.y[i] = ys[i];
  }
  int nbins = 0;
  for (int i = 0; i < nscales; ++i) nbins += oh[i] * ow[i];
  MR_CHECK_ARG(nbins <= MR_POOL_MULTI_BINS, "mr_adaptive_avgpool_multi_fwd: %d bins > %d", nbins, MR_POOL_MULTI_BINS);
  const int cpb = dtype == MR_F32 ? 32 : 64;
  const int lds = H * W * 128 + nbins * 8;
  DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_avgpool_multi_fwd_kernel<T>), dim3(cdiv(C, cpb), N), dim3(256), lds, stream,
                                       (const T*)x, a, N, H, W, C, nbins));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_adaptive_avgpool_multi_bwd(int dtype, void* const* dys, const int* oh, const int* ow, int nscales, void* dx, int N, int H,
                                  int W, int C, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(nscales > 0 && nscales <= MR_POOL_MULTI_MAX && C % vec == 0, "mr_adaptive_avgpool_multi_bwd: bad arguments");
  PoolMultiArgs a;
  a.nsc = nscales;
  for (int i = 0; i < nscales; ++i) {
    MR_CHECK_ARG(dys[i] != nullptr && oh[i] > 0 && ow[i] > 0, "mr_adaptive_avgpool_multi_bwd: bad scale %d", i);
    a.oh[i] = oh[i]; a.ow[i] = ow[i]; a.y[i] = dys[i];
  }
  int nbins = 0;
  for (int i = 0; i < nscales; ++i) nbins += oh[i] * ow[i];
  MR_CHECK_ARG(nbins <= MR_POOL_MULTI_BINS, "mr_adaptive_avgpool_multi_bwd: %d bins > %d", nbins, MR_POOL_MULTI_BINS);
  const int cpb = dtype == MR_F32 ? 32 : 64;
  const int lds = nbins * (128 + 4) + nscales * (H + W) * 4;
  MR_CHECK_ARG(lds <= 65536, "mr_adaptive_avgpool_multi_bwd: tables (%d B) too large for LDS", lds);
  DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_avgpool_multi_bwd_kernel<T>), dim3(cdiv(C, cpb), N), dim3(256), lds, stream,
                                       a, (T*)dx, N, H, W, C, nbins));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_bilinear_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, int OH, int OW, int ldy, int coff,
                    int accumulate, hipStream_t stream) {
  MR_CHECK_ARG(ldy >= coff + C, "mr_bilinear_fwd: channel slice out of range");
  const int vec = dtype == MR_F32 ? 4 : 8;
  if (C % vec == 0 && ldy % vec == 0 && coff % vec == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_fwd_vec_kernel<T>), dim3(grid_for((long long)N * OH * OW * (C / vec), 256)),
                                         dim3(256), 0, stream, (const T*)x, (T*)y, N, H, W, C, OH, OW, ldy, coff, accumulate));
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  const long long total = (long long)N * OH * OW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)x, (T*)y, N, H, W, C, OH, OW, ldy, coff, accumulate));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_bilinear_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int OH, int OW, int lddy,
                    int coff, hipStream_t stream) {
  MR_CHECK_ARG(lddy >= coff + C, "mr_bilinear_bwd: channel slice out of range");
  const int vec = dtype == MR_F32 ? 4 : 8;
  if (C % vec == 0 && lddy % vec == 0 && coff % vec == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_bwd_vec_kernel<T>), dim3(grid_for((long long)N * H * W * (C / vec), 256)),
                                         dim3(256), 0, stream, (const T*)dy, (T*)dx, N, H, W, C, OH, OW, lddy, coff));
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  const long long total = (long long)N * H * W * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)dy, (T*)dx, N, H, W, C, OH, OW, lddy, coff));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_copy_channels(int dtype, const void* src, int lds, int soff, void* dst, int ldd, int doff, long long P, int C,
                     hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0 && lds % vec == 0 && ldd % vec == 0 && soff % vec == 0 && doff % vec == 0,
               "mr_copy_channels: sizes/offsets must be multiples of %d", vec);
  MR_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "mr_copy_channels: 16-byte alignment");
  DISPATCH_T(dtype, hipLaunchKernelGGL((copy_channels_kernel<T>), dim3(grid_for(P * (C / vec), 256)), dim3(256), 0,
                                       stream, (const T*)src, lds, soff, (T*)dst, ldd, doff, P, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_scatter_strided(int dtype, const void* dxs, void* dx, int N, int H, int W, int C, int sh, int sw, int Ho, int Wo,
                       hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0 && sh >= 1 && sw >= 1 && Ho >= 1 && Wo >= 1 &&
                   (Ho - 1) * sh < H && (Wo - 1) * sw < W,
               "mr_scatter_strided: bad shape N=%d H=%d W=%d C=%d stride %dx%d Ho=%d Wo=%d", N, H, W, C, sh, sw, Ho, Wo);
  MR_CHECK_ARG(((uintptr_t)dxs & 15) == 0 && ((uintptr_t)dx & 15) == 0, "mr_scatter_strided: 16-byte alignment");
  DISPATCH_T(dtype, hipLaunchKernelGGL((scatter_strided_kernel<T>), dim3(grid_for((long long)N * H * W * (C / vec), 256)),
                                       dim3(256), 0, stream, (const T*)dxs, (T*)dx, N, H, W, C, sh, sw, Ho, Wo));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_scale_channels(int dtype, const void* x, const float* scale, void* y, int N, long long HW, int C,
                      hipStream_t stream) {
  const long long total = (long long)N * HW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((scale_channels_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)x, scale, (T*)y, N, HW, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_ctc2d_head_fwd(int dtype, const void* mask_logits, int lda, const void* cls_logits, int ldz, float* lp,
                      float* mask_prob, float* cls_prob, int N, int H, int W, int C, float tiny, hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0, "mr_ctc2d_head_fwd: bad shape");
  DISPATCH_T(dtype, hipLaunchKernelGGL((ctc2d_head_fwd_kernel<T>), dim3(cdiv(N * W, 4)), dim3(256), 0, stream,
                                       (const T*)mask_logits, lda, (const T*)cls_logits, ldz, lp, mask_prob,
                                       cls_prob, N, H, W, C, tiny));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_ctc2d_head_bwd(int dtype, const float* grad_lp, const float* mask_prob, const float* cls_prob, void* dmask,
                      int ldda, void* dcls, int lddz, int N, int H, int W, int C, float tiny, hipStream_t stream) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((ctc2d_head_bwd_kernel<T>), dim3(cdiv(N * W, 4)), dim3(256), 0, stream,
                                       grad_lp, mask_prob, cls_prob, (T*)dmask, ldda, (T*)dcls, lddz, N, H, W, C,
                                       tiny));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// nearest upsampling x [N,H,W,C] -> y [N,H*s,W*s, ld = ldy] channel slice [coff, coff+C); add (nullable) [N,H*s,W*s,C]
int mr_nearest_up_fwd(int dtype, const void* x, const void* add, void* y, int N, int H, int W, int C, int s, int ldy,
                      int coff, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(s >= 1 && C % vec == 0 && ldy % vec == 0 && coff % vec == 0 && ldy >= coff + C,
               "mr_nearest_up_fwd: bad shape C=%d ldy=%d coff=%d s=%d", C, ldy, coff, s);
  const long long total = (long long)N * H * s * W * s * (C / vec);
  DISPATCH_T(dtype, hipLaunchKernelGGL((nearest_up_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)x, (const T*)add, (T*)y, N, H, W, C, s, ldy, coff));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_deconv2x2_d2s(int dtype, const void* y2, int ld2, const float* bias, void* y, int N, int H, int W, int C,
                     hipStream_t stream) {
  MR_CHECK_ARG(y2 && y && N > 0 && H > 0 && W > 0 && C > 0 && ld2 >= 4 * C, "mr_deconv2x2_d2s: bad arguments");
  const int vec = dtype == MR_F32 ? 4 : 8;
  const long long total = (long long)N * H * W * ((C + vec - 1) / vec);
  DISPATCH_T(dtype, hipLaunchKernelGGL((deconv2x2_d2s_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)y2, ld2, bias, (T*)y, N, H, W, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_deconv2x2_s2d(int dtype, const void* dy, void* dy2, int ld2, int N, int H, int W, int C, hipStream_t stream) {
  MR_CHECK_ARG(dy && dy2 && N > 0 && H > 0 && W > 0 && C > 0 && ld2 >= 4 * C, "mr_deconv2x2_s2d: bad arguments");
  const int vec = dtype == MR_F32 ? 4 : 8;
  const long long total = (long long)N * H * W * ((C + vec - 1) / vec);
  DISPATCH_T(dtype, hipLaunchKernelGGL((deconv2x2_s2d_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)dy, (T*)dy2, ld2, N, H, W, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_nearest_up_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int s, int lddy, int coff,
                      hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(s >= 1 && C % vec == 0 && lddy % vec == 0 && coff % vec == 0 && lddy >= coff + C,
               "mr_nearest_up_bwd: bad shape C=%d lddy=%d coff=%d s=%d", C, lddy, coff, s);
  const long long total = (long long)N * H * W * (C / vec);
  DISPATCH_T(dtype, hipLaunchKernelGGL((nearest_up_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)dy, (T*)dx, N, H, W, C, s, lddy, coff));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
