// Geometry + sampling descriptor shared by the DCNv2 kernels (dcn.hip: general fallback; dcn_fused.hip: fused kernels).
// Sampling rule of the reference (assets/ops/dcn/src/deform_conv_cuda_kernel.cu:466-496, 569-632): tap (i, j) of output
// pixel (ho, wo) samples x at (ho*stride - pad + i*dil + dh, wo*stride - pad + j*dil + dw); the sample counts iff
// h > -1 && w > -1 && h < H && w < W; bilinear over the four integer neighbours, neighbours outside the image = 0;
// offsets / mask are FLAT [2*taps][Ho][Wo] / [taps][Ho][Wo] f32 arrays from the base of the sample's buffer (quirk Q10).
#pragma once
#include "common.h"

namespace mr {

struct DcnGeom {
  int N, H, W, C, Ho, Wo, kh, kw, stride, pad, dil;
  long long off_bs, msk_bs;  // per-sample strides (elements) of the offset / mask buffers
};

__device__ __forceinline__ bool dcn_point(const DcnGeom& g, const float* off_b, int tap, int ho, int wo, float& ph,
                                          float& pw) {
  const int i = tap / g.kw, j = tap - i * g.kw;
  const long long o = ((long long)(2 * tap) * g.Ho + ho) * g.Wo + wo;
  ph = (float)(ho * g.stride - g.pad + i * g.dil) + off_b[o];
  pw = (float)(wo * g.stride - g.pad + j * g.dil) + off_b[o + (long long)g.Ho * g.Wo];
  return ph > -1.f && pw > -1.f && ph < (float)g.H && pw < (float)g.W;
}

// One (output pixel, tap) sample: fractional parts, modulation mask and the top-left integer neighbour.
// An invalid sample has m = 0 and (hl, wl) = (-2, -2): none of its four neighbours is inside the image.
struct DcnDesc {
  float lh, lw, m;
  int hl, wl;
};

__device__ __forceinline__ DcnDesc dcn_desc_invalid() {
  DcnDesc d;
  d.m = 0.f; d.lh = 0.f; d.lw = 0.f; d.hl = -2; d.wl = -2;
  return d;
}

// descriptor from already-loaded offset (dh, dw) and mask values -- lets a kernel fetch the three floats a pipeline stage
// ahead of the corner loads that depend on them
__device__ __forceinline__ DcnDesc dcn_desc_from(const DcnGeom& g, int tap, int ho, int wo, float off_h, float off_w,
                                                 float m) {
  const int i = tap / g.kw, j = tap - i * g.kw;
  const float ph = (float)(ho * g.stride - g.pad + i * g.dil) + off_h;
  const float pw = (float)(wo * g.stride - g.pad + j * g.dil) + off_w;
  if (!(ph > -1.f && pw > -1.f && ph < (float)g.H && pw < (float)g.W)) return dcn_desc_invalid();
  DcnDesc d;
  const float fh = floorf(ph), fw = floorf(pw);
  d.m = m;
  d.hl = (int)fh;
  d.wl = (int)fw;
  d.lh = ph - fh;
  d.lw = pw - fw;
  return d;
}

__device__ __forceinline__ DcnDesc dcn_desc(const DcnGeom& g, const float* __restrict__ offset,
                                            const float* __restrict__ mask, int n, int tap, int ho, int wo) {
  const long long hw = (long long)g.Ho * g.Wo, o = (long long)ho * g.Wo + wo;
  const float* ob = offset + n * g.off_bs + (2 * tap) * hw + o;
  return dcn_desc_from(g, tap, ho, wo, ob[0], ob[hw], mask[n * g.msk_bs + tap * hw + o]);
}

// packed 16-byte form for LDS tables
__device__ __forceinline__ float4 dcn_pack(const DcnDesc& d) {
  const int hw = (d.hl & 0xffff) | (int)((unsigned)d.wl << 16);
  return make_float4(d.lh, d.lw, d.m, __int_as_float(hw));
}
__device__ __forceinline__ DcnDesc dcn_unpack(const float4& v) {
  DcnDesc d;
  d.lh = v.x; d.lw = v.y; d.m = v.z;
  const int hw = __float_as_int(v.w);
  d.hl = (int)(short)(hw & 0xffff);
  d.wl = hw >> 16;
  return d;
}

__device__ __forceinline__ bool dcn_inside(const DcnGeom& g, int h, int w) {
  return (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
}

}  // namespace mr
