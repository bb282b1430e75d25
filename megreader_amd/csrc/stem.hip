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
constexpr int STEM_BWD_GROUPS = 512;  // workgroups (= partial sums) of the backward main kernel

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

// dw[o][c][dr][ds] (strided) += sum_g partial[g][k][o];  dbias[o] += sum_g partial[g][9*CIN][o].  One block per k.
template <int CIN>
__global__ __launch_bounds__(512) void stem_bwd_reduce_kernel(const float* __restrict__ partial, int G,
                                                              float* __restrict__ dw, long long dsk, long long dsc,
                                                              long long dsr, long long dss,
                                                              float* __restrict__ dbias) {
  constexpr int KT = 9 * CIN;
  __shared__ float red[8][STEM_COUT];
  const int k = blockIdx.x, o = threadIdx.x & 63, part = threadIdx.x >> 6;  // 8 parts
  float s = 0.f;
  for (int g = part; g < G; g += 8) s += partial[((long long)g * (KT + 1) + k) * STEM_COUT + o];
  red[part][o] = s;
  __syncthreads();
  if (part == 0) {
#pragma unroll
    for (int p = 1; p < 8; ++p) s += red[p][o];
    if (k < KT) {
      const int c = k % CIN, t = k / CIN, ds = t % 3, dr = t / 3;
      if (dw) dw[o * dsk + c * dsc + dr * dsr + ds * dss] += s;
    } else if (dbias) {
      dbias[o] += s;
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

int mr_stem_fwd(int dtype, const float* x, const float* w, long long wsk, long long wsc, long long wsr,
                long long wss, const float* bias, void* y, unsigned char* code, int N, int Cin, int H, int W,
                hipStream_t stream) {
  if (int rc = stem_check("mr_stem_fwd", N, Cin, H, W)) return rc;
  const size_t lds = sizeof(float) * 4 * (W + 2) * Cin;
  const int strips = N * (H / 2);
  const int grid = strips < 8192 ? strips : 8192;
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
  if (Cin == 3) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((stem_bwd_kernel<T, 3>), dim3(G), dim3(256), lds, stream, (const T*)dy, code,
                                         x, workspace, N, H, W));
    hipLaunchKernelGGL((stem_bwd_reduce_kernel<3>), dim3(9 * 3 + 1), dim3(512), 0, stream, workspace, G, dw, dsk, dsc,
                       dsr, dss, dbias);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((stem_bwd_kernel<T, 1>), dim3(G), dim3(256), lds, stream, (const T*)dy, code,
                                         x, workspace, N, H, W));
    hipLaunchKernelGGL((stem_bwd_reduce_kernel<1>), dim3(9 * 1 + 1), dim3(512), 0, stream, workspace, G, dw, dsk, dsc,
                       dsr, dss, dbias);
  }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
