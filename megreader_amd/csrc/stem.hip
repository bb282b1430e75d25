// Fused first stage of the CRNN backbone: Conv2d(Cin -> 64, 3x3, stride 1, pad 1) + bias + ReLU + MaxPool2d(2, 2).
// (reference backbones/crnn.py:17-19 + 48-55: cnn.conv0 / relu0 / pooling0 on the 3-channel 32x128 crop.)
//
// Why a dedicated kernel: with Cin = 3 the layer is pure HBM traffic.  As separate ops it writes the 64-channel
// full-resolution activation (134 MB at batch 256), reads it back to pool, and in backward writes / re-reads the
// equally large sparse gradient.  Fused, the forward reads the fp32 NCHW image (12.6 MB) and writes only the pooled
// activation (33.5 MB) plus a one-byte code per pooled element; the backward reads the pooled gradient, the codes
// and the image, and produces dW / db directly -- the full-resolution tensors never exist.
//
// Mapping: one wavefront lane = one output channel (Cout == 64 == wavefront width), so each lane keeps its 9*Cin
// filter taps (forward) or 9*Cin + 1 gradient accumulators (backward) in registers for the whole kernel.  A
// workgroup owns one pooled output row ("strip"): the 4 input rows it needs are staged in LDS once, already rounded
// to the compute dtype, and the 4x4xCin patch of each pooled pixel is read from LDS with wave-uniform addresses
// (hardware broadcast, no bank conflicts).  Arithmetic is f32 FMA on compute-dtype-rounded operands: the same
// products the MFMA path forms, summed in a different order.
//
// Code byte per pooled element: bits 0-1 = position of the first maximum in the 2x2 window (row-major, the order
// nn.MaxPool2d scans), bit 2 = the maximum is > 0 (ReLU passes gradient).
#include "common.h"
#include "../../include/megreader_hip.h"

namespace mr {

constexpr int STEM_COUT = 64;
constexpr int STEM_BWD_GROUPS = 1024;  // workgroups (= partial sums) of the backward main kernel

template <typename T> __device__ __forceinline__ float round_as(float v) { return to_f32(from_f32<T>(v)); }

// patch[r][col + 1][c], r = 0..3 <-> input rows 2*ph - 1 + r, col = -1..W (zero outside the image)
template <typename T, int CIN>
__device__ __forceinline__ void stem_stage_patch(float* patch, const float* __restrict__ x, int n, int ph, int H,
                                                 int W) {
  const int PW = W + 2;
  for (int r = 0; r < 4; ++r) {
    const int row = 2 * ph - 1 + r;
    const bool rv = row >= 0 && row < H;
    for (int j = threadIdx.x; j < CIN * W; j += blockDim.x) {
      const int c = j / W, col = j - c * W;
      const float v = rv ? x[((long long)(n * CIN + c) * H + row) * W + col] : 0.f;
      patch[(r * PW + col + 1) * CIN + c] = round_as<T>(v);
    }
  }
  if (threadIdx.x < 4 * 2 * CIN) {
    const int r = threadIdx.x / (2 * CIN), rem = threadIdx.x - r * 2 * CIN;
    const int side = rem / CIN, c = rem - side * CIN;
    patch[(r * PW + (side ? W + 1 : 0)) * CIN + c] = 0.f;
  }
}

template <typename T, int CIN>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       long long wsk, long long wsc, long long wsr, long long wss,
                                                       const float* __restrict__ bias, T* __restrict__ y,
                                                       unsigned char* __restrict__ code, int N, int H, int W) {
  extern __shared__ float patch[];
  constexpr int KT = 9 * CIN;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const int Ho = H / 2, Wo = W / 2, PW = W + 2;
  float wr[KT];  // [dr][ds][c]
#pragma unroll
  for (int dr = 0; dr < 3; ++dr)
#pragma unroll
    for (int ds = 0; ds < 3; ++ds)
#pragma unroll
      for (int c = 0; c < CIN; ++c)
        wr[(dr * 3 + ds) * CIN + c] = round_as<T>(w[lane * wsk + c * wsc + dr * wsr + ds * wss]);
  const float b = bias ? bias[lane] : 0.f;
  const int per = (Wo + nwv - 1) / nwv;
  for (int strip = blockIdx.x; strip < N * Ho; strip += gridDim.x) {
    const int n = strip / Ho, ph = strip - n * Ho;
    __syncthreads();
    stem_stage_patch<T, CIN>(patch, x, n, ph, H, W);
    __syncthreads();
    const int pw1 = min(Wo, (wv + 1) * per);
    for (int pw = wv * per; pw < pw1; ++pw) {
      float P[4][4 * CIN];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4 * CIN; ++i) P[r][i] = patch[(r * PW + 2 * pw) * CIN + i];
      float best = 0.f;
      int q = 0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int a = qq >> 1, bq = qq & 1;
        float acc = b;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr)
#pragma unroll
          for (int i = 0; i < 3 * CIN; ++i) acc = fmaf(wr[dr * 3 * CIN + i], P[a + dr][bq * CIN + i], acc);
        if (qq == 0 || acc > best) {  // strict > keeps the first maximum
          best = acc;
          q = qq;
        }
      }
      const bool pos = best > 0.f;
      const long long o = ((long long)strip * Wo + pw) * STEM_COUT + lane;
      y[o] = from_f32<T>(pos ? best : 0.f);
      code[o] = (unsigned char)(q | (pos ? 4 : 0));
    }
  }
}

// partial[g][k][o], k = 0..9*CIN-1 filter taps ([dr][ds][c]) and k = 9*CIN the bias gradient
template <typename T, int CIN>
__global__ __launch_bounds__(256) void stem_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ code,
                                                       const float* __restrict__ x, float* __restrict__ partial, int N,
                                                       int H, int W) {
  extern __shared__ float patch[];
  constexpr int KT = 9 * CIN;
  __shared__ float red[4][KT + 1][STEM_COUT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const int Ho = H / 2, Wo = W / 2, PW = W + 2;
  float acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = 0.f;
  float accb = 0.f;
  const int per = (Wo + nwv - 1) / nwv;
  for (int strip = blockIdx.x; strip < N * Ho; strip += gridDim.x) {
    const int n = strip / Ho, ph = strip - n * Ho;
    __syncthreads();
    stem_stage_patch<T, CIN>(patch, x, n, ph, H, W);
    __syncthreads();
    const int pw1 = min(Wo, (wv + 1) * per);
    for (int pw = wv * per; pw < pw1; ++pw) {
      const long long o = ((long long)strip * Wo + pw) * STEM_COUT + lane;
      const int cd = code[o];
      const float g = (cd & 4) ? to_f32(dy[o]) : 0.f;
      const int q = cd & 3;
      float P[4][4 * CIN];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4 * CIN; ++i) P[r][i] = patch[(r * PW + 2 * pw) * CIN + i];
      accb += g;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int a = qq >> 1, bq = qq & 1;
        const float gq = (q == qq) ? g : 0.f;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr)
#pragma unroll
          for (int i = 0; i < 3 * CIN; ++i)
            acc[dr * 3 * CIN + i] = fmaf(gq, P[a + dr][bq * CIN + i], acc[dr * 3 * CIN + i]);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KT; ++k) red[wv][k][lane] = acc[k];
  red[wv][KT][lane] = accb;
  __syncthreads();
  for (int i = threadIdx.x; i < (KT + 1) * STEM_COUT; i += blockDim.x) {
    const int k = i / STEM_COUT, o = i - k * STEM_COUT;
    float s = 0.f;
    for (int v = 0; v < nwv; ++v) s += red[v][k][o];
    partial[(long long)blockIdx.x * (KT + 1) * STEM_COUT + i] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 MFMA variants (Wo % 16 == 0).  The 3x3xCin filter is a K = 9*Cin <= 27 (padded to 32) reduction: exactly one
// v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 channels.  Operand fragments are gathered from the bf16 LDS patch:
// k = (dr*3 + ds)*Cin + c lives at patch offset dr*ROW + (ds*Cin + c) from the pixel's top-left element.
// ------------------------------------------------------------------------------------------------------------------
// Division-free staging: waves 0,2 fetch patch rows 0,1 and waves 1,3 rows 2,3; a wave pair covers 128 columns per
// pass.  All global loads of a thread are issued before its first LDS store.
template <int CIN>
__device__ __forceinline__ void stem_stage_patch_bf16(bf16_t* patch, const float* __restrict__ x, int n, int ph,
                                                      int H, int W) {
  const int PW = W + 2;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rbase = (wv & 1) * 2;
  for (int col = lane + 64 * (wv >> 1); col < W; col += 128) {
    float v[2][CIN];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * ph - 1 + rbase + rr;
      const bool rv = row >= 0 && row < H;
#pragma unroll
      for (int c = 0; c < CIN; ++c) v[rr][c] = rv ? x[((long long)(n * CIN + c) * H + row) * W + col] : 0.f;
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int c = 0; c < CIN; ++c) patch[((rbase + rr) * PW + col + 1) * CIN + c] = (bf16_t)v[rr][c];
  }
  if (threadIdx.x < 4 * 2 * CIN) {
    const int r = threadIdx.x / (2 * CIN), rem = threadIdx.x - r * 2 * CIN;
    const int side = rem / CIN, c = rem - side * CIN;
    patch[(r * PW + (side ? W + 1 : 0)) * CIN + c] = (bf16_t)0.f;
  }
}

// Filter bank packed for the MFMA forward: bf16 [64 channels][32 k], k = (dr*3 + ds)*Cin + c, zero for k >= 9*Cin.
// Produced once per optimizer step (MR_PREP_STEM job of mr_prep_batch, or mr_stem_pack).
template <int CIN>
__global__ void stem_pack_kernel(const float* __restrict__ w, long long wsk, long long wsc, long long wsr,
                                 long long wss, bf16_t* __restrict__ wpack) {
  constexpr int KT = 9 * CIN;
  for (int i = threadIdx.x; i < STEM_COUT * 32; i += blockDim.x) {
    const int ch = i >> 5, k = i & 31;
    const int dr = k / (3 * CIN), rem = k - dr * 3 * CIN, ds = rem / CIN, c = rem - ds * CIN;
    wpack[i] = (bf16_t)(k < KT ? w[ch * wsk + c * wsc + dr * wsr + ds * wss] : 0.f);
  }
}

// Forward.  One workgroup per pooled output row (strip); wave = 16 pooled pixels: 4 window positions x 4 channel
// tiles = 16 MFMAs.  MFMA row r of channel tile ct is channel (r>>2)*16 + ct*4 + (r&3), so that a lane (pixel l15,
// row group lg) ends up with the 16 CONSECUTIVE channels lg*16 .. lg*16+15 of its pixel for all four window
// positions: max / arg-max / ReLU happen in registers and the lane writes 32 contiguous bytes of y and 16 of codes.
// The kernel is VALU-bound (arg-max bookkeeping), so everything else is kept off the vector ALU: the filter bank
// arrives pre-packed (4 fragment loads per lane), staging is division-free.
template <int CIN>
__global__ __launch_bounds__(256, 3) void stem_fwd_mfma_kernel(const float* __restrict__ x,
                                                            const bf16_t* __restrict__ wpack,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                            unsigned char* __restrict__ code, int N, int H, int W) {
  extern __shared__ float patch_raw[];
  bf16_t* patch = (bf16_t*)patch_raw;
  constexpr int KT = 9 * CIN;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
  const int Ho = H / 2, Wo = W / 2, PW = W + 2, ROW = PW * CIN;
  bf16x8 wf[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int ch = (l15 >> 2) * 16 + ct * 4 + (l15 & 3);
    wf[ct] = *(const bf16x8*)(wpack + ch * 32 + 8 * lg);
  }
  f32x4 bv[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
    bv[ct] = bias ? *(const f32x4*)(bias + lg * 16 + ct * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  int koff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = 8 * lg + i;
    const int dr = k / (3 * CIN), rem = k - dr * 3 * CIN;
    koff[i] = k < KT ? dr * ROW + rem : 0;
  }

  for (int strip = blockIdx.x; strip < N * Ho; strip += gridDim.x) {
    const int n = strip / Ho, ph = strip - n * Ho;
    __syncthreads();
    stem_stage_patch_bf16<CIN>(patch, x, n, ph, H, W);
    __syncthreads();
    for (int pw0 = wv * 16; pw0 < Wo; pw0 += 64) {
      const int pw = pw0 + l15;
      bf16x8 xf[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int base = ((qq >> 1) * PW + 2 * pw + (qq & 1)) * CIN;
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[qq][i] = patch[base + koff[i]];
      }
      bf16_t yo[16];
      unsigned char co[16];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        f32x4 acc[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          acc[qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], xf[qq], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // first maximum of the window, in scan order
          // (compare + select instead of fmaxf: no NaN canonicalisation, and the compares are needed anyway)
          const float a0 = acc[0][r], a1 = acc[1][r], a2 = acc[2][r], a3 = acc[3][r];
          const bool c1 = a1 > a0, c2 = a3 > a2;
          const float m01 = c1 ? a1 : a0, m23 = c2 ? a3 : a2;
          const bool c3 = m23 > m01;
          const int q = c3 ? (c2 ? 3 : 2) : (c1 ? 1 : 0);
          const float best = (c3 ? m23 : m01) + bv[ct][r];
          const bool pos = best > 0.f;
          yo[ct * 4 + r] = (bf16_t)(pos ? best : 0.f);
          co[ct * 4 + r] = (unsigned char)(pos ? (q | 4) : q);
        }
      }
      const long long o = ((long long)strip * Wo + pw) * STEM_COUT + lg * 16;
      *(uint4*)(y + o) = *(const uint4*)&yo[0];
      *(uint4*)(y + o + 8) = *(const uint4*)&yo[8];
      *(uint4*)(code + o) = *(const uint4*)&co[0];
    }
  }
}

// Backward.  dW[ch][k] = sum over (pooled pixel, window position) of G[ch] * X[k]: per 8 pooled pixels one
// 32-deep reduction step (8 pixels x 4 positions).  Lane group lg owns pixels 2*lg, 2*lg+1 and all 4 positions, so
// the G fragment of a lane (channel l15 of tile mt) needs just 2 gradient values + 2 codes.  An extra im2col column
// k = 9*Cin of ones yields the bias gradient in the same MFMAs.  The gradient / code loads of a chunk are issued
// one chunk ahead (the first one before the patch barrier) so they overlap the staging and the MFMAs.  Fragments
// are assembled with 32-bit integer ops on the raw bf16 bits (the kernel is VALU-bound).
template <int CIN>
__global__ __launch_bounds__(256, 4) void stem_bwd_mfma_kernel(const bf16_t* __restrict__ dy,
                                                            const unsigned char* __restrict__ code,
                                                            const float* __restrict__ x, float* __restrict__ partial,
                                                            int N, int H, int W) {
  extern __shared__ float patch_raw[];
  bf16_t* patch = (bf16_t*)patch_raw;
  const unsigned short* patch16 = (const unsigned short*)patch_raw;
  const unsigned short* dy16 = (const unsigned short*)dy;
  constexpr int KT = 9 * CIN;
  __shared__ float red[4][KT + 1][STEM_COUT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
  const int Ho = H / 2, Wo = W / 2, PW = W + 2, ROW = PW * CIN;
  // X-fragment gather offsets of this lane: k = l15 + 16*nt, reduction slot i -> pixel 2*lg + (i>>2), position i&3.
  // Lanes of the bias column (k == 9*Cin) use all-ones, lanes beyond it zeros: value = (raw & xand) | xor_.
  int xoff[2][8];
  unsigned xand[2], xor_[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int k = l15 + 16 * nt;
    const int dr = k / (3 * CIN), rem = k - dr * 3 * CIN;
    xand[nt] = k < KT ? 0xFFFFFFFFu : 0u;
    xor_[nt] = k == KT ? 0x3F803F80u : 0u;  // bf16 1.0 in both halves
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int px = 2 * lg + (i >> 2), qq = i & 3;
      xoff[nt][i] = k < KT ? ((qq >> 1) * PW + 2 * px + (qq & 1)) * CIN + dr * ROW + rem : 0;
    }
  }
  f32x4 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned gd[4][2], gc[4][2];
  auto load_chunk = [&](int strip, int pw0) {
    const long long o = ((long long)strip * Wo + pw0 + 2 * lg) * STEM_COUT + l15;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const long long oo = o + (long long)h * STEM_COUT + mt * 16;
        gd[mt][h] = dy16[oo];
        gc[mt][h] = code[oo];
      }
  };

  for (int strip = blockIdx.x; strip < N * Ho; strip += gridDim.x) {
    const int n = strip / Ho, ph = strip - n * Ho;
    if (wv * 8 < Wo) load_chunk(strip, wv * 8);
    __syncthreads();
    stem_stage_patch_bf16<CIN>(patch, x, n, ph, H, W);
    __syncthreads();
    for (int pw0 = wv * 8; pw0 < Wo; pw0 += 32) {
      // G fragment: dword d of (mt) holds reduction slots 2d, 2d+1 = pixel h = d>>1, positions 2(d&1), 2(d&1)+1
      u32x4 gf[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned cd = gc[mt][h];
          const unsigned g = (cd & 4u) ? gd[mt][h] : 0u;
          const unsigned sh = g << ((cd & 1u) * 16u);   // low half for even positions, high half for odd
          gf[mt][h * 2 + 0] = (cd & 2u) ? 0u : sh;     // positions 0,1
          gf[mt][h * 2 + 1] = (cd & 2u) ? sh : 0u;     // positions 2,3
        }
      if (pw0 + 32 < Wo) load_chunk(strip, pw0 + 32);  // next chunk's loads fly during this chunk's MFMAs
      u32x4 xf[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned lo = patch16[2 * pw0 * CIN + xoff[nt][2 * d]];
          const unsigned hi = patch16[2 * pw0 * CIN + xoff[nt][2 * d + 1]];
          xf[nt][d] = ((lo | (hi << 16)) & xand[nt]) | xor_[nt];
        }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&gf[mt], *(const bf16x8*)&xf[nt],
                                                                acc[mt][nt], 0, 0, 0);
    }
  }
  __syncthreads();
  // D rows = channels mt*16 + lg*4 + r, columns = k = l15 + 16*nt
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int k = l15 + 16 * nt;
      if (k <= KT)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][k][mt * 16 + lg * 4 + r] = acc[mt][nt][r];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < (KT + 1) * STEM_COUT; i += blockDim.x) {
    const int k = i / STEM_COUT, oc = i - k * STEM_COUT;
    partial[(long long)blockIdx.x * (KT + 1) * STEM_COUT + i] = red[0][k][oc] + red[1][k][oc] + red[2][k][oc] + red[3][k][oc];
  }
}

// dw[o][c][dr][ds] (strided) += sum_g partial[g][k][o];  dbias[o] += sum_g partial[g][9*CIN][o].
// grid (9*CIN + 1, 8): block (k, z) sums the partials g = z, z+8, ... with 8 thread groups, then one atomic per (k, o).
template <int CIN>
__global__ __launch_bounds__(512) void stem_bwd_reduce_kernel(const float* __restrict__ partial, int G,
                                                              float* __restrict__ dw, long long dsk, long long dsc,
                                                              long long dsr, long long dss,
                                                              float* __restrict__ dbias) {
  constexpr int KT = 9 * CIN;
  __shared__ float red[8][STEM_COUT];
  const int k = blockIdx.x, o = threadIdx.x & 63, part = threadIdx.x >> 6;  // 8 parts
  float s = 0.f;
  for (int g = blockIdx.y + 8 * part; g < G; g += 64) s += partial[((long long)g * (KT + 1) + k) * STEM_COUT + o];
  red[part][o] = s;
  __syncthreads();
  if (part == 0) {
#pragma unroll
    for (int p = 1; p < 8; ++p) s += red[p][o];
    if (k < KT) {
      const int c = k % CIN, t = k / CIN, ds = t % 3, dr = t / 3;
      if (dw) atomicAdd(&dw[o * dsk + c * dsc + dr * dsr + ds * dss], s);
    } else if (dbias) {
      atomicAdd(&dbias[o], s);
    }
  }
}

}  // namespace mr

using namespace mr;

#define DISPATCH_T(dtype, ...)                                   \
  if ((dtype) == MR_F32) { typedef float T; __VA_ARGS__; }       \
  else if ((dtype) == MR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { mr::set_error("bad dtype %d", (dtype)); return MR_ERR_DTYPE; }

static int stem_check(const char* who, int N, int Cin, int H, int W) {
  if (!(N > 0 && (Cin == 1 || Cin == 3) && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 &&
        4ll * (W + 2) * Cin * 4 <= 48 * 1024)) {
    mr::set_error("%s: unsupported shape N=%d Cin=%d H=%d W=%d (Cin in {1,3}, even H and W, W <= ~1000)", who, N,
                  Cin, H, W);
    return MR_ERR_ARG;
  }
  return MR_OK;
}

extern "C" {

long long mr_stem_bwd_workspace(int Cin) { return (long long)STEM_BWD_GROUPS * (9 * Cin + 1) * STEM_COUT; }

// bf16 filter bank for the MFMA forward path ([64][32], see stem_pack_kernel)
int mr_stem_pack(const float* w, long long wsk, long long wsc, long long wsr, long long wss, void* wpack, int Cin,
                 hipStream_t stream) {
  MR_CHECK_ARG(Cin == 1 || Cin == 3, "mr_stem_pack: Cin must be 1 or 3");
  if (Cin == 3)
    hipLaunchKernelGGL((stem_pack_kernel<3>), dim3(1), dim3(256), 0, stream, w, wsk, wsc, wsr, wss, (bf16_t*)wpack);
  else
    hipLaunchKernelGGL((stem_pack_kernel<1>), dim3(1), dim3(256), 0, stream, w, wsk, wsc, wsr, wss, (bf16_t*)wpack);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_stem_fwd(int dtype, const float* x, const float* w, long long wsk, long long wsc, long long wsr,
                long long wss, const void* wpack, const float* bias, void* y, unsigned char* code, int N, int Cin,
                int H, int W, hipStream_t stream) {
  if (int rc = stem_check("mr_stem_fwd", N, Cin, H, W)) return rc;
  const size_t lds = sizeof(float) * 4 * (W + 2) * Cin;
  const int strips = N * (H / 2);
  const int grid = strips < 8192 ? strips : 8192;
  if (dtype == MR_BF16 && (W / 2) % 16 == 0 && wpack != nullptr) {  // MFMA path
    if (Cin == 3)
      hipLaunchKernelGGL((stem_fwd_mfma_kernel<3>), dim3(grid), dim3(256), lds, stream, x, (const bf16_t*)wpack, bias,
                         (bf16_t*)y, code, N, H, W);
    else
      hipLaunchKernelGGL((stem_fwd_mfma_kernel<1>), dim3(grid), dim3(256), lds, stream, x, (const bf16_t*)wpack, bias,
                         (bf16_t*)y, code, N, H, W);
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  if (Cin == 3) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((stem_fwd_kernel<T, 3>), dim3(grid), dim3(256), lds, stream, x, w, wsk, wsc,
                                         wsr, wss, bias, (T*)y, code, N, H, W));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((stem_fwd_kernel<T, 1>), dim3(grid), dim3(256), lds, stream, x, w, wsk, wsc,
                                         wsr, wss, bias, (T*)y, code, N, H, W));
  }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// dw / dbias are ACCUMULATED into (either may be null); workspace: mr_stem_bwd_workspace(Cin) floats.
int mr_stem_bwd(int dtype, const void* dy, const unsigned char* code, const float* x, float* workspace, float* dw,
                long long dsk, long long dsc, long long dsr, long long dss, float* dbias, int N, int Cin, int H,
                int W, hipStream_t stream) {
  if (int rc = stem_check("mr_stem_bwd", N, Cin, H, W)) return rc;
  MR_CHECK_ARG(workspace != nullptr, "mr_stem_bwd: workspace is null");
  const size_t lds = sizeof(float) * 4 * (W + 2) * Cin;
  const int strips = N * (H / 2);
  const int G = strips < STEM_BWD_GROUPS ? strips : STEM_BWD_GROUPS;
  if (dtype == MR_BF16 && (W / 2) % 16 == 0) {  // MFMA path
    if (Cin == 3) {
      hipLaunchKernelGGL((stem_bwd_mfma_kernel<3>), dim3(G), dim3(256), lds, stream, (const bf16_t*)dy, code, x,
                         workspace, N, H, W);
      hipLaunchKernelGGL((stem_bwd_reduce_kernel<3>), dim3(9 * 3 + 1, 8), dim3(512), 0, stream, workspace, G, dw, dsk,
                         dsc, dsr, dss, dbias);
    } else {
      hipLaunchKernelGGL((stem_bwd_mfma_kernel<1>), dim3(G), dim3(256), lds, stream, (const bf16_t*)dy, code, x,
                         workspace, N, H, W);
      hipLaunchKernelGGL((stem_bwd_reduce_kernel<1>), dim3(9 * 1 + 1, 8), dim3(512), 0, stream, workspace, G, dw, dsk,
                         dsc, dsr, dss, dbias);
    }
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  if (Cin == 3) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((stem_bwd_kernel<T, 3>), dim3(G), dim3(256), lds, stream, (const T*)dy, code,
                                         x, workspace, N, H, W));
    hipLaunchKernelGGL((stem_bwd_reduce_kernel<3>), dim3(9 * 3 + 1, 8), dim3(512), 0, stream, workspace, G, dw, dsk, dsc,
                       dsr, dss, dbias);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((stem_bwd_kernel<T, 1>), dim3(G), dim3(256), lds, stream, (const T*)dy, code,
                                         x, workspace, N, H, W));
    hipLaunchKernelGGL((stem_bwd_reduce_kernel<1>), dim3(9 * 1 + 1, 8), dim3(512), 0, stream, workspace, G, dw, dsk, dsc,
                       dsr, dss, dbias);
  }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
