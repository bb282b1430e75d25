// Modulated deformable convolution v2 (DCNv2) sampling kernels for gfx950, NHWC activations.
// Replaces assets/ops/dcn/src/deform_conv_cuda_kernel.cu:569-766 (modulated im2col / col2im / col2im_coord) and,
// together with the MFMA GEMMs (mr_gemm_nt / mr_gemm_tn), the host functions deform_conv_cuda.cpp:486-679.
//
// Differences from the reference by design: the whole batch is processed in one launch and one GEMM (the reference
// loops over images, one im2col + one cuBLAS GEMM each); activations are channel-contiguous, so every bilinear
// corner is one 16-byte vector load shared by 8 channels; the column matrix is [pixels][tap*C + c] = the A operand
// of the NT GEMM against KRSC weights, in the compute dtype.  Offsets / masks are read exactly like the reference
// does: per sample, as FLAT [2*9][Ho][Wo] / [9][Ho][Wo] f32 arrays from the base of that sample's (possibly
// larger) NCHW buffer (reference quirk Q10) -- their gradients are written back the same way.
// All three kernels are HBM / L2-bound gather-scatter work; groups = deformable_groups = 1.
#include "dcn_geom.h"
#include "../../include/megreader_hip.h"

namespace mr {

// fused kernels (dcn_fused.hip): no column matrix, no floating-point atomics on the input gradient
bool dcn_fused_ok(int dtype, int H, int W, int C, int Co, int kh, int kw);
long long dcn_fused_ws_bytes(int dtype, int N, int H, int W, int C, int Ho, int Wo, int taps);
int dcn_fused_fwd(int dtype, const void* x, const void* w_n, const float* bias, const float* offset, const float* mask,
                  void* y, void* ws, const DcnGeom& g, int Co, hipStream_t stream);
long long dcn_fused_fwd_ws_bytes(int dtype, int C, int N, int Ho, int Wo, int Co, int taps);
bool dcn_use_col_fwd(int dtype, int C);
bool dcn_fused_dx_direct(int dtype, int N, int H, int W, int C, int taps);
int dcn_fused_bwd(int dtype, const void* dy, const void* x, const void* w_t, const float* offset, const float* mask,
                  void* ws, float* dx32, void* dx_t, int flags, float* doffset, float* dmask, float* dw, float* dbias,
                  const DcnGeom& g, int Co, const void* col_saved, hipStream_t stream);

// col[p, tap*C + c] = valid ? mask * bilinear(x[n,:,:,c], p_tap) : 0     thread = (p, tap, 16-byte channel vector)
template <typename T>
__global__ void dcn2_im2col_kernel(const T* __restrict__ x, const float* __restrict__ offset,
                                   const float* __restrict__ mask, T* __restrict__ col, DcnGeom g) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = g.C / VEC;
  const int taps = g.kh * g.kw;
  const long long total = (long long)g.N * g.Ho * g.Wo * taps * cv;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(t % cv) * VEC;
    long long q = t / cv;
    const int tap = (int)(q % taps);
    const long long p = q / taps;
    const int wo = (int)(p % g.Wo);
    const long long r = p / g.Wo;
    const int ho = (int)(r % g.Ho);
    const int n = (int)(r / g.Ho);
    float ph, pw;
    const bool valid = dcn_point(g, offset + n * g.off_bs, tap, ho, wo, ph, pw);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (valid) {
      const float m = mask[n * g.msk_bs + ((long long)tap * g.Ho + ho) * g.Wo + wo];
      const int hl = (int)floorf(ph), wl = (int)floorf(pw);
      const float lh = ph - (float)hl, lw = pw - (float)wl;
      const int hh = hl + 1, wh = wl + 1;
      const T* xb = x + (long long)n * g.H * g.W * g.C + c0;
      const float wgt[4] = {(1.f - lh) * (1.f - lw) * m, (1.f - lh) * lw * m, lh * (1.f - lw) * m, lh * lw * m};
      const int hs[4] = {hl, hl, hh, hh}, ws[4] = {wl, wh, wl, wh};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (hs[k] < 0 || hs[k] > g.H - 1 || ws[k] < 0 || ws[k] > g.W - 1) continue;
        const uint4 v = *(const uint4*)(xb + ((long long)hs[k] * g.W + ws[k]) * g.C);
        const T* pv = (const T*)&v;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += wgt[k] * to_f32(pv[j]);
      }
    }
    uint4 out;
    T* po = (T*)&out;
#pragma unroll
    for (int j = 0; j < VEC; ++j) po[j] = from_f32<T>(acc[j]);
    *(uint4*)(col + p * ((long long)taps * g.C) + (long long)tap * g.C + c0) = out;
  }
}

// gradient w.r.t. offsets and mask: one wave per (p, tap), lanes stride the channels, wave reduction.
//   dmask[tap]      = sum_c gcol * (valid ? bilinear : 0)
//   doff[2tap+dir]  = sum_c gcol * mask * d bilinear / d p_dir   (0 when invalid)
template <typename T>
__global__ void dcn2_coord_kernel(const T* __restrict__ gcol, const T* __restrict__ x,
                                  const float* __restrict__ offset, const float* __restrict__ mask,
                                  float* __restrict__ doffset, float* __restrict__ dmask, DcnGeom g) {
  const int taps = g.kh * g.kw;
  const int lane = threadIdx.x & 63;
  const long long item = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long items = (long long)g.N * g.Ho * g.Wo * taps;
  if (item >= items) return;
  const int tap = (int)(item % taps);
  const long long p = item / taps;
  const int wo = (int)(p % g.Wo);
  const long long r = p / g.Wo;
  const int ho = (int)(r % g.Ho);
  const int n = (int)(r / g.Ho);
  float ph, pw;
  const bool valid = dcn_point(g, offset + n * g.off_bs, tap, ho, wo, ph, pw);
  float dm = 0.f, dh = 0.f, dw = 0.f;
  if (valid) {
    const float m = mask[n * g.msk_bs + ((long long)tap * g.Ho + ho) * g.Wo + wo];
    const int hl = (int)floorf(ph), wl = (int)floorf(pw);
    const float lh = ph - (float)hl, lw = pw - (float)wl;
    const int hh = hl + 1, wh = wl + 1;
    const bool o1 = hl >= 0 && wl >= 0, o2 = hl >= 0 && wh <= g.W - 1, o3 = hh <= g.H - 1 && wl >= 0,
               o4 = hh <= g.H - 1 && wh <= g.W - 1;
    const T* xb = x + (long long)n * g.H * g.W * g.C;
    const T* gp = gcol + p * ((long long)taps * g.C) + (long long)tap * g.C;
    for (int c = lane; c < g.C; c += 64) {
      const float gv = to_f32(gp[c]);
      const float v1 = o1 ? to_f32(xb[((long long)hl * g.W + wl) * g.C + c]) : 0.f;
      const float v2 = o2 ? to_f32(xb[((long long)hl * g.W + wh) * g.C + c]) : 0.f;
      const float v3 = o3 ? to_f32(xb[((long long)hh * g.W + wl) * g.C + c]) : 0.f;
      const float v4 = o4 ? to_f32(xb[((long long)hh * g.W + wh) * g.C + c]) : 0.f;
      dm += gv * ((1.f - lh) * (1.f - lw) * v1 + (1.f - lh) * lw * v2 + lh * (1.f - lw) * v3 + lh * lw * v4);
      dh += gv * m * (-(1.f - lw) * v1 - lw * v2 + (1.f - lw) * v3 + lw * v4);
      dw += gv * m * (-(1.f - lh) * v1 + (1.f - lh) * v2 - lh * v3 + lh * v4);
    }
  }
  dm = wave_sum(dm);
  dh = wave_sum(dh);
  dw = wave_sum(dw);
  if (lane == 0) {
    const long long hw = (long long)g.Ho * g.Wo, o = (long long)ho * g.Wo + wo;
    doffset[n * g.off_bs + (2 * tap) * hw + o] = dh;
    doffset[n * g.off_bs + (2 * tap + 1) * hw + o] = dw;
    dmask[n * g.msk_bs + tap * hw + o] = dm;
  }
}

// gradient w.r.t. the input: dx[n, corner, c] += gcol * mask * bilinear weight   (f32 atomics; 4 corners)
template <typename T>
__global__ void dcn2_col2im_kernel(const T* __restrict__ gcol, const float* __restrict__ offset,
                                   const float* __restrict__ mask, float* __restrict__ dx, DcnGeom g) {
  const int taps = g.kh * g.kw;
  const long long total = (long long)g.N * g.Ho * g.Wo * taps * g.C;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(t % g.C);
    long long q = t / g.C;
    const int tap = (int)(q % taps);
    const long long p = q / taps;
    const int wo = (int)(p % g.Wo);
    const long long r = p / g.Wo;
    const int ho = (int)(r % g.Ho);
    const int n = (int)(r / g.Ho);
    float ph, pw;
    if (!dcn_point(g, offset + n * g.off_bs, tap, ho, wo, ph, pw)) continue;
    const float m = mask[n * g.msk_bs + ((long long)tap * g.Ho + ho) * g.Wo + wo];
    const float gv = to_f32(gcol[p * ((long long)taps * g.C) + (long long)tap * g.C + c]) * m;
    if (gv == 0.f) continue;
    const int hl = (int)floorf(ph), wl = (int)floorf(pw);
    const float lh = ph - (float)hl, lw = pw - (float)wl;
    const int hh = hl + 1, wh = wl + 1;
    float* db = dx + (long long)n * g.H * g.W * g.C + c;
    if (hl >= 0 && wl >= 0) atomicAdd(db + ((long long)hl * g.W + wl) * g.C, gv * (1.f - lh) * (1.f - lw));
    if (hl >= 0 && wh <= g.W - 1) atomicAdd(db + ((long long)hl * g.W + wh) * g.C, gv * (1.f - lh) * lw);
    if (hh <= g.H - 1 && wl >= 0) atomicAdd(db + ((long long)hh * g.W + wl) * g.C, gv * lh * (1.f - lw));
    if (hh <= g.H - 1 && wh <= g.W - 1) atomicAdd(db + ((long long)hh * g.W + wh) * g.C, gv * lh * lw);
  }
}

// ---- round-2 backward kernel experiments (off by default: measured slower / equal, see mr_tuning.dcn_v1_bwd) -----------
// coord v2: 8 lanes per (pixel, tap) item, each lane a 16-byte channel vector of gcol and of the four corners (the
// first version put one WAVE on an item with 2-byte loads per lane: 295 us per layer at batch 16).  The three sums are
// reduced over the 8 lanes with three shuffle steps.  C must be a multiple of 8 * VEC (64 for bf16).
template <typename T>
__global__ __launch_bounds__(256) void dcn2_coord_vec_kernel(const T* __restrict__ gcol, const T* __restrict__ x,
                                                              const float* __restrict__ offset,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ doffset, float* __restrict__ dmask,
                                                              DcnGeom g) {
  constexpr int VEC = VecOf<T>::N;
  const int taps = g.kh * g.kw;
  const int lv = threadIdx.x & 7;
  const long long item = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 3;
  const long long items = (long long)g.N * g.Ho * g.Wo * taps;
  const bool live = item < items;
  const long long it = live ? item : 0;
  const int tap = (int)(it % taps);
  const long long p = it / taps;
  const int wo = (int)(p % g.Wo);
  const long long r = p / g.Wo;
  const int ho = (int)(r % g.Ho);
  const int n = (int)(r / g.Ho);
  float ph, pw;
  const bool valid = live && dcn_point(g, offset + n * g.off_bs, tap, ho, wo, ph, pw);
  float dm = 0.f, dh = 0.f, dw = 0.f;
  if (valid) {
    const float m = mask[n * g.msk_bs + ((long long)tap * g.Ho + ho) * g.Wo + wo];
    const int hl = (int)floorf(ph), wl = (int)floorf(pw);
    const float lh = ph - (float)hl, lw = pw - (float)wl;
    const int hh = hl + 1, wh = wl + 1;
    const bool o1 = hl >= 0 && wl >= 0, o2 = hl >= 0 && wh <= g.W - 1, o3 = hh <= g.H - 1 && wl >= 0,
               o4 = hh <= g.H - 1 && wh <= g.W - 1;
    const float a1 = (1.f - lh) * (1.f - lw), a2 = (1.f - lh) * lw, a3 = lh * (1.f - lw), a4 = lh * lw;
    const T* xb = x + (long long)n * g.H * g.W * g.C;
    const T* gp = gcol + p * ((long long)taps * g.C) + (long long)tap * g.C;
    const T* c1 = xb + ((long long)hl * g.W + wl) * g.C;
    const T* c2 = xb + ((long long)hl * g.W + wh) * g.C;
    const T* c3 = xb + ((long long)hh * g.W + wl) * g.C;
    const T* c4 = xb + ((long long)hh * g.W + wh) * g.C;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int c = lv * VEC; c < g.C; c += 8 * VEC) {
      const uint4 gq = *(const uint4*)(gp + c);
      const uint4 q1 = o1 ? *(const uint4*)(c1 + c) : z, q2 = o2 ? *(const uint4*)(c2 + c) : z;
      const uint4 q3 = o3 ? *(const uint4*)(c3 + c) : z, q4 = o4 ? *(const uint4*)(c4 + c) : z;
      const T* pg = (const T*)&gq;
      const T *p1 = (const T*)&q1, *p2 = (const T*)&q2, *p3 = (const T*)&q3, *p4 = (const T*)&q4;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float gv = to_f32(pg[j]);
        const float v1 = to_f32(p1[j]), v2 = to_f32(p2[j]), v3 = to_f32(p3[j]), v4 = to_f32(p4[j]);
        dm += gv * (a1 * v1 + a2 * v2 + a3 * v3 + a4 * v4);
        dh += gv * m * (-(1.f - lw) * v1 - lw * v2 + (1.f - lw) * v3 + lw * v4);
        dw += gv * m * (-(1.f - lh) * v1 + (1.f - lh) * v2 - lh * v3 + lh * v4);
      }
    }
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    dm += __shfl_xor(dm, o, 64);
    dh += __shfl_xor(dh, o, 64);
    dw += __shfl_xor(dw, o, 64);
  }
  if (live && lv == 0) {
    const long long hw = (long long)g.Ho * g.Wo, o = (long long)ho * g.Wo + wo;
    doffset[n * g.off_bs + (2 * tap) * hw + o] = dh;
    doffset[n * g.off_bs + (2 * tap + 1) * hw + o] = dw;
    dmask[n * g.msk_bs + tap * hw + o] = dm;
  }
}

// col2im v2: LDS pre-reduction.  The first version issued one f32 global atomic per (pixel, tap, channel, corner):
// 36 atomics land on every input element (9 taps x 4 corners) -- 814 us per layer at batch 16, 55 % of the DCN backward.
// A workgroup now owns a TH x TW tile of output pixels of one image and a chunk of CC channels, accumulates all their
// contributions into an LDS image of the input patch the tile can reach with offsets up to +-R pixels (ds_add_f32, lanes
// of one item on distinct banks), and flushes the patch with ONE coalesced global atomic per touched element.  Samples
// that fall outside the patch (offsets beyond R) go straight to global memory as before, so any offset stays correct.
constexpr int C2I_TH = 8, C2I_TW = 8, C2I_R = 2;
template <typename T>
__global__ __launch_bounds__(256) void dcn2_col2im_lds_kernel(const T* __restrict__ gcol,
                                                               const float* __restrict__ offset,
                                                               const float* __restrict__ mask, float* __restrict__ dx,
                                                               DcnGeom g, int CC, int PH, int PW, int tiles_h,
                                                               int tiles_w) {
  constexpr int VEC = VecOf<T>::N;
  extern __shared__ float c2i_patch[];   // [PH * PW][CC + 1]: the odd pixel stride spreads different pixels over the banks
  const int PS = CC + 1;                 // (with stride CC every item of a wave hit the same 8 banks: 8-way conflicts)
  const int taps = g.kh * g.kw;
  const int cchunks = g.C / CC;
  int b = blockIdx.x;
  const int cc = b % cchunks; b /= cchunks;
  const int tw = b % tiles_w; b /= tiles_w;
  const int th = b % tiles_h;
  const int n = b / tiles_h;
  const int ho0 = th * C2I_TH, wo0 = tw * C2I_TW;
  const int h0 = ho0 * g.stride - g.pad - C2I_R, w0 = wo0 * g.stride - g.pad - C2I_R;   // patch origin (input coords)
  const int npatch = PH * PW * PS;
  for (int i = threadIdx.x; i < npatch; i += 256) c2i_patch[i] = 0.f;
  __syncthreads();
  const int cvv = CC / VEC;                      // channel vectors of this chunk per (pixel, tap)
  const int nitems = C2I_TH * C2I_TW * taps * cvv;
  const float* off_b = offset + n * g.off_bs;
  float* dxn = dx + (long long)n * g.H * g.W * g.C + cc * CC;
  for (int it = threadIdx.x; it < nitems; it += 256) {
    const int lv = it % cvv;
    int q = it / cvv;
    const int tap = q % taps; q /= taps;
    const int tx = q % C2I_TW, ty = q / C2I_TW;
    const int ho = ho0 + ty, wo = wo0 + tx;
    if (ho >= g.Ho || wo >= g.Wo) continue;
    float ph, pw;
    if (!dcn_point(g, off_b, tap, ho, wo, ph, pw)) continue;
    const float m = mask[n * g.msk_bs + ((long long)tap * g.Ho + ho) * g.Wo + wo];
    const long long p = ((long long)n * g.Ho + ho) * g.Wo + wo;
    const uint4 gq = *(const uint4*)(gcol + p * ((long long)taps * g.C) + (long long)tap * g.C + cc * CC + lv * VEC);
    const T* pg = (const T*)&gq;
    const int hl = (int)floorf(ph), wl = (int)floorf(pw);
    const float lh = ph - (float)hl, lw = pw - (float)wl;
    const int hh = hl + 1, wh = wl + 1;
    const float wgt[4] = {(1.f - lh) * (1.f - lw) * m, (1.f - lh) * lw * m, lh * (1.f - lw) * m, lh * lw * m};
    const int ch[4] = {hl, hl, hh, hh}, cw[4] = {wl, wh, wl, wh};
    const bool inpatch = hl >= h0 && hh < h0 + PH && wl >= w0 && wh < w0 + PW;
    // lanes lv and lv + 4 of an item would share banks: the upper four walk their channels rotated by half a vector.
    // Both candidate indices are compile-time constants and a select picks one: a run-time index into the register
    // vector (pg[(j + rot) & 7]) sent it to SCRATCH memory -- 32 scratch loads per item, 1.46 ms per layer.
    const bool hi = (lv & 4) != 0;
    float val[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) val[j] = to_f32(pg[j]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (ch[k] < 0 || ch[k] > g.H - 1 || cw[k] < 0 || cw[k] > g.W - 1) continue;   // corner outside the image
      if (inpatch) {
        float* d = c2i_patch + ((ch[k] - h0) * PW + (cw[k] - w0)) * PS + lv * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          constexpr int HALF = VEC / 2;
          const int j1 = (j + HALF) & (VEC - 1);
          atomicAdd(d + (hi ? j1 : j), (hi ? val[j1] : val[j]) * wgt[k]);
        }
      } else {
        float* d = dxn + ((long long)ch[k] * g.W + cw[k]) * g.C + lv * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) atomicAdd(d + j, val[j] * wgt[k]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < npatch; i += 256) {
    const float v = c2i_patch[i];
    if (v == 0.f) continue;
    const int c = i % PS, px = (i / PS) % PW, py = i / (PS * PW);
    const int h = h0 + py, w = w0 + px;
    if (c >= CC || h < 0 || h > g.H - 1 || w < 0 || w > g.W - 1) continue;
    atomicAdd(dxn + ((long long)h * g.W + w) * g.C + c, v);
  }
}

#define g_dcn_v1_bwd MR_TUNE(dcn_v1_bwd)   // 1 (default): round-1 backward kernels; 0: the round-2 experiments (mr_tuning.dcn_v1_bwd)

static inline int grid_for(long long n, int block, int max_blocks = 32768) {
  long long b = (n + block - 1) / block;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

static int make_geom(DcnGeom& g, int N, int H, int W, int C, int kh, int kw, int stride, int pad, int dil, int Ho,
                     int Wo, long long off_bs, long long msk_bs) {
  g.N = N; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo; g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad;
  g.dil = dil; g.off_bs = off_bs; g.msk_bs = msk_bs;
  MR_CHECK_ARG(Ho == (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1 &&
                   Wo == (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1,
               "dcn: output size %dx%d inconsistent with geometry", Ho, Wo);
  MR_CHECK_ARG(off_bs >= (long long)2 * kh * kw * Ho * Wo && msk_bs >= (long long)kh * kw * Ho * Wo,
               "dcn: offset / mask buffers are smaller than the output grid needs");
  return MR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Packed offset/mask operand (how the reference's deformable ResNet feeds the op, backbones/resnet.py:125-142:
// offset_mask = conv2_offset(x); offset = offset_mask[:, :18]; mask = offset_mask[:, -9:].sigmoid()).  The offset conv's
// output lives here as NHWC [N][HW][ld] in the compute dtype (27 channels padded to ld); the sampling kernels read flat f32
// NCHW offsets / masks.  One launch each way replaces the slice / cast / contiguous / sigmoid (forward: 5 launches per
// layer) and sigmoid_backward / slice_backward / add / re-pad (backward: ~10 launches per layer) of the unfused graph.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void dcn_unpack_kernel(const T* __restrict__ raw, int ld, float* __restrict__ off, float* __restrict__ msk,
                                  int N, int HW, int noff, int nmsk) {
  const long long total = (long long)N * HW;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(t / HW), p = (int)(t % HW);
    const T* r = raw + t * ld;
    float* o = off + (long long)n * noff * HW + p;
    for (int c = 0; c < noff; ++c) o[(long long)c * HW] = to_f32(r[c]);
    float* m = msk + (long long)n * nmsk * HW + p;
    for (int c = 0; c < nmsk; ++c) m[(long long)c * HW] = 1.f / (1.f + expf(-to_f32(r[noff + c])));
  }
}

// graw[n][p][c] = doff[n][c][p] (c < noff) | dmsk[n][c-noff][p] * m * (1 - m) (c < noff + nmsk) | 0 (padding up to ld)
template <typename T>
__global__ void dcn_pack_grad_kernel(const float* __restrict__ doff, const float* __restrict__ dmsk,
                                     const float* __restrict__ msk, T* __restrict__ graw, int ld, int N, int HW,
                                     int noff, int nmsk) {
  const long long total = (long long)N * HW;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(t / HW), p = (int)(t % HW);
    T* r = graw + t * ld;
    const float* o = doff + (long long)n * noff * HW + p;
    for (int c = 0; c < noff; ++c) r[c] = from_f32<T>(o[(long long)c * HW]);
    const float* gm = dmsk + (long long)n * nmsk * HW + p;
    const float* m = msk + (long long)n * nmsk * HW + p;
    for (int c = 0; c < nmsk; ++c) {
      const float mv = m[(long long)c * HW];
      r[noff + c] = from_f32<T>(gm[(long long)c * HW] * mv * (1.f - mv));
    }
    for (int c = noff + nmsk; c < ld; ++c) r[c] = from_f32<T>(0.f);
  }
}

}  // namespace mr

using namespace mr;

#define DISPATCH_T(dtype, ...)                                   \
  if ((dtype) == MR_F32) { typedef float T; __VA_ARGS__; }       \
  else if ((dtype) == MR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { mr::set_error("bad dtype %d", (dtype)); return MR_ERR_DTYPE; }

extern "C" {

// x NHWC [N,H,W,C] (`dtype`); offset f32 [N][>= 2*kh*kw*Ho*Wo] / mask f32 [N][>= kh*kw*Ho*Wo] with per-sample
// strides off_bs / msk_bs; col [N*Ho*Wo, kh*kw*C] (`dtype`)
int mr_dcn2_im2col(int dtype, const void* x, const float* offset, long long off_bs, const float* mask,
                   long long msk_bs, void* col, int N, int H, int W, int C, int kh, int kw, int stride, int pad,
                   int dil, int Ho, int Wo, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_dcn2_im2col: C (%d) must be a multiple of %d", C, vec);
  DcnGeom g;
  int rc = make_geom(g, N, H, W, C, kh, kw, stride, pad, dil, Ho, Wo, off_bs, msk_bs);
  if (rc) return rc;
  const long long total = (long long)N * Ho * Wo * kh * kw * (C / vec);
  DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_im2col_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)x, offset, mask, (T*)col, g));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// doffset / dmask: f32 buffers shaped like offset / mask (the entries addressed by the flat [.,Ho,Wo] view are
// written, the rest must be pre-zeroed by the caller like the reference's torch.zeros_like)
int mr_dcn2_coord_grad(int dtype, const void* gcol, const void* x, const float* offset, long long off_bs,
                       const float* mask, long long msk_bs, float* doffset, float* dmask, int N, int H, int W,
                       int C, int kh, int kw, int stride, int pad, int dil, int Ho, int Wo, hipStream_t stream) {
  DcnGeom g;
  int rc = make_geom(g, N, H, W, C, kh, kw, stride, pad, dil, Ho, Wo, off_bs, msk_bs);
  if (rc) return rc;
  const long long items = (long long)N * Ho * Wo * kh * kw;
  const int vec = dtype == MR_F32 ? 4 : 8;
  if (C % (8 * vec) == 0 && !g_dcn_v1_bwd) {   // 8 lanes x one 16-byte vector per item
    DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_coord_vec_kernel<T>), dim3((unsigned)cdivll(items * 8, 256)), dim3(256), 0,
                                         stream, (const T*)gcol, (const T*)x, offset, mask, doffset, dmask, g));
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_coord_kernel<T>), dim3((unsigned)cdivll(items, 4)), dim3(256), 0, stream,
                                       (const T*)gcol, (const T*)x, offset, mask, doffset, dmask, g));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// dx: f32 NHWC [N,H,W,C], pre-zeroed, accumulated with atomics
int mr_dcn2_col2im(int dtype, const void* gcol, const float* offset, long long off_bs, const float* mask,
                   long long msk_bs, float* dx, int N, int H, int W, int C, int kh, int kw, int stride, int pad,
                   int dil, int Ho, int Wo, hipStream_t stream) {
  DcnGeom g;
  int rc = make_geom(g, N, H, W, C, kh, kw, stride, pad, dil, Ho, Wo, off_bs, msk_bs);
  if (rc) return rc;
  const long long total = (long long)N * Ho * Wo * kh * kw * C;
  const int vec = dtype == MR_F32 ? 4 : 8;
  // LDS-tiled form: patch of the inputs an 8 x 8 output tile reaches with offsets up to +-R, CC channels, <= 64 KB
  const int PH = (C2I_TH - 1) * stride + (kh - 1) * dil + 2 + 2 * C2I_R;
  const int PW = (C2I_TW - 1) * stride + (kw - 1) * dil + 2 + 2 * C2I_R;
  int CC = 64;
  while (CC > vec && (long long)PH * PW * (CC + 1) * 4 > 64 * 1024) CC >>= 1;
  if (!g_dcn_v1_bwd && C % CC == 0 && CC % vec == 0 && (long long)PH * PW * (CC + 1) * 4 <= 64 * 1024) {
    const int tiles_h = cdiv(Ho, C2I_TH), tiles_w = cdiv(Wo, C2I_TW);
    const long long blocks = (long long)N * tiles_h * tiles_w * (C / CC);
    MR_CHECK_ARG(blocks < (1ll << 31), "mr_dcn2_col2im: grid too large");
    const size_t lds = (size_t)PH * PW * (CC + 1) * 4;
    DISPATCH_T(dtype, {
      auto kern = dcn2_col2im_lds_kernel<T>;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, (const T*)gcol, offset, mask, dx, g, CC,
                         PH, PW, tiles_h, tiles_w);
    });
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_col2im_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)gcol, offset, mask, dx, g));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// A/B: 1 (default) = the round-1 backward kernels (one wave per coordinate item, one global atomic per corner and
// channel); 0 = the round-2 experiments.  MEASURED (tools/microbench_dcn.py, 13 layers, batch 16): LDS-tiled col2im
// 1460 us per layer vs 830 us for direct global atomics -- ds_add_f32 sustains well under one lane-operation per clock
// per CU on gfx950 (the same cliff as the LDS-atomic column sums of the wide-tile TN kernel), independent of bank
// conflicts (stride 64 vs 65 floats: same time); vectorised coordinate gradient 302 vs 295 us (both bound by the 36x
// re-read of the input corners from L2, not by the access width).
// bytes of the caller-owned workspace `col_ws` of mr_dcn2_fwd (backward = 0) / mr_dcn2_bwd (backward = 1) for this shape:
// fused path: nothing forward, the CSR of the scatter pattern backward; general path: the column matrix.
long long mr_dcn2_ws_bytes(int dtype, int N, int H, int W, int C, int Co, int kh, int kw, int Ho, int Wo, int backward) {
  if (dcn_fused_ok(dtype, H, W, C, Co, kh, kw))
    return backward ? dcn_fused_ws_bytes(dtype, N, H, W, C, Ho, Wo, kh * kw) : dcn_fused_fwd_ws_bytes(dtype, C, N, Ho, Wo, Co, kh * kw);
  return (long long)N * Ho * Wo * kh * kw * C * (dtype == MR_F32 ? 4 : 2);
}

// ---- single-call forms: what `modulated_deform_conv_cuda_forward / _backward`
// (assets/ops/dcn/src/deform_conv_cuda.cpp:486-679) are to the reference's python Function.  The caller owns every
// buffer, including the column workspace (the reference passes `columns` the same way, functions/deform_conv.py:135).
// x / y / dy NHWC in `dtype`; w_n [Co][kh*kw*C] and w_t [kh*kw*C][Co] in `dtype` (mr_prep_matrix images of the KRSC
// weight); col_ws [N*Ho*Wo, kh*kw*C] in `dtype`.
int mr_dcn2_fwd(int dtype, const void* x, const void* w_n, const float* bias, const float* offset, long long off_bs,
                const float* mask, long long msk_bs, void* y, void* col_ws, int N, int H, int W, int C, int Co, int kh,
                int kw, int stride, int pad, int dil, int Ho, int Wo, hipStream_t stream) {
  if (dcn_fused_ok(dtype, H, W, C, Co, kh, kw)) {   // sample -> LDS -> MFMA, no column matrix (col_ws = f32 accumulator of the tap-split launch for small layers, else unused)
    DcnGeom g;
    int rcg = make_geom(g, N, H, W, C, kh, kw, stride, pad, dil, Ho, Wo, off_bs, msk_bs);
    if (rcg) return rcg;
    return dcn_fused_fwd(dtype, x, w_n, bias, offset, mask, y, col_ws, g, Co, stream);
  }
  MR_CHECK_ARG(col_ws != nullptr, "mr_dcn2_fwd: this shape needs a column workspace (mr_dcn2_ws_bytes)");
  int rc = mr_dcn2_im2col(dtype, x, offset, off_bs, mask, msk_bs, col_ws, N, H, W, C, kh, kw, stride, pad, dil, Ho, Wo,
                          stream);
  if (rc) return rc;
  const int K = kh * kw * C;
  return mr_gemm_nt(dtype, col_ws, K, w_n, K, y, Co, bias, 0, N * Ho * Wo, Co, K, stream);
}

// dx32 (nullable): f32 NHWC, pre-zeroed, accumulated; doffset / dmask: f32, shaped like offset / mask, pre-zeroed;
// dw (nullable): f32 [Co][kh*kw*C] accumulated; dbias (nullable): f32 [Co] accumulated.  col_ws is used twice (dy * W,
// then the recomputed column matrix for dW), exactly like `columns` in deform_conv_cuda.cpp:611-665.
int mr_dcn2_bwd(int dtype, const void* dy, const void* x, const void* w_t, const float* offset, long long off_bs,
                const float* mask, long long msk_bs, void* col_ws, float* dx32, float* doffset, float* dmask, float* dw,
                float* dbias, int N, int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int Ho,
                int Wo, hipStream_t stream) {
  return mr_dcn2_bwd2(dtype, dy, x, w_t, offset, off_bs, mask, msk_bs, col_ws, dx32, nullptr, 0, doffset, dmask, dw, dbias, N, H,
                      W, C, Co, kh, kw, stride, pad, dil, Ho, Wo, stream);
}

// host only: 1 when the fused kernels (dcn_fused.hip) serve this shape under the current mr_tuning.dcn_fused
int mr_dcn2_fused(int dtype, int H, int W, int C, int Co, int kh, int kw) { return dcn_fused_ok(dtype, H, W, C, Co, kh, kw) ? 1 : 0; }

// host only: 1 when mr_dcn2_bwd2 can write the input gradient directly in the compute dtype (dx_t) for this shape
int mr_dcn2_dx_direct(int dtype, int N, int H, int W, int C, int Co, int kh, int kw) {
  return (dcn_fused_ok(dtype, H, W, C, Co, kh, kw) && dcn_fused_dx_direct(dtype, N, H, W, C, kh * kw)) ? 1 : 0;
}

// mr_dcn2_bwd with two launch-saving options of the fused path (round 5; the 13 DCN layers of the batch-2 detector are ~10
// launch-floor-sized launches each): dx_t (nullable, instead of dx32) = the input gradient in `dtype`, OVERWRITTEN -- no zero
// fill in front, no conversion pass behind (only where mr_dcn2_dx_direct says 1); flags bit 0 = the workspace is one that was
// zeroed ONCE and has since been used by this function only (its counters return to zero by themselves): no memset node.
int mr_dcn2_bwd2(int dtype, const void* dy, const void* x, const void* w_t, const float* offset, long long off_bs,
                 const float* mask, long long msk_bs, void* col_ws, float* dx32, void* dx_t, int flags, float* doffset,
                 float* dmask, float* dw, float* dbias, int N, int H, int W, int C, int Co, int kh, int kw, int stride, int pad,
                 int dil, int Ho, int Wo, hipStream_t stream) {
  return mr_dcn2_bwd3(dtype, dy, x, w_t, offset, off_bs, mask, msk_bs, col_ws, dx32, dx_t, flags, doffset, dmask, dw, dbias, nullptr,
                      N, H, W, C, Co, kh, kw, stride, pad, dil, Ho, Wo, stream);
}

// host only: 1 when mr_dcn2_fwd leaves the sampled column matrix [N*Ho*Wo, kh*kw*C] of this layer in its col_ws (bf16 shapes of the
// materialised path under mr_tuning.dcn_col_fwd): a caller that keeps that buffer alive hands it to mr_dcn2_bwd3 as col_saved
int mr_dcn2_col_saved(int dtype, int H, int W, int C, int Co, int kh, int kw) {
  return (dcn_fused_ok(dtype, H, W, C, Co, kh, kw) && dcn_use_col_fwd(dtype, C)) ? 1 : 0;
}

// mr_dcn2_bwd2 + col_saved (nullable): the column matrix the forward wrote (mr_dcn2_col_saved) -- the weight gradient then runs
// straight off it instead of sampling x again (round 6).  Ignored by the general path and by float32.
int mr_dcn2_bwd3(int dtype, const void* dy, const void* x, const void* w_t, const float* offset, long long off_bs,
                 const float* mask, long long msk_bs, void* col_ws, float* dx32, void* dx_t, int flags, float* doffset,
                 float* dmask, float* dw, float* dbias, const void* col_saved, int N, int H, int W, int C, int Co, int kh, int kw,
                 int stride, int pad, int dil, int Ho, int Wo, hipStream_t stream) {
  if (dcn_fused_ok(dtype, H, W, C, Co, kh, kw)) {
    DcnGeom g;
    int rcg = make_geom(g, N, H, W, C, kh, kw, stride, pad, dil, Ho, Wo, off_bs, msk_bs);
    if (rcg) return rcg;
    return dcn_fused_bwd(dtype, dy, x, w_t, offset, mask, col_ws, dx32, dx_t, flags, doffset, dmask, dw, dbias, g, Co,
                         dcn_use_col_fwd(dtype, C) ? col_saved : nullptr, stream);
  }
  MR_CHECK_ARG(dx_t == nullptr, "mr_dcn2_bwd2: dx_t needs the fused path (mr_dcn2_dx_direct)");
  MR_CHECK_ARG(col_ws != nullptr, "mr_dcn2_bwd: workspace missing (mr_dcn2_ws_bytes)");
  const int K = kh * kw * C, P = N * Ho * Wo;
  int rc = MR_OK;
  if ((doffset && dmask) || dx32) {
    rc = mr_gemm_nt(dtype, dy, Co, w_t, Co, col_ws, K, nullptr, 0, P, K, Co, stream);   // gcol = dy * W
    if (rc) return rc;
  }
  if (doffset && dmask) {
    rc = mr_dcn2_coord_grad(dtype, col_ws, x, offset, off_bs, mask, msk_bs, doffset, dmask, N, H, W, C, kh, kw, stride,
                            pad, dil, Ho, Wo, stream);
    if (rc) return rc;
  }
  if (dx32) {
    rc = mr_dcn2_col2im(dtype, col_ws, offset, off_bs, mask, msk_bs, dx32, N, H, W, C, kh, kw, stride, pad, dil, Ho, Wo,
                        stream);
    if (rc) return rc;
  }
  if (dw || dbias) {
    rc = mr_dcn2_im2col(dtype, x, offset, off_bs, mask, msk_bs, col_ws, N, H, W, C, kh, kw, stride, pad, dil, Ho, Wo,
                        stream);
    if (rc) return rc;
    if (dw)
      rc = mr_gemm_tn(dtype, dy, Co, col_ws, K, dw, K, P, Co, K, 0, dbias, stream);
    else
      rc = mr_colsum(dtype, dy, dbias, P, Co, Co, 0, stream);
  }
  return rc;
}

int mr_dcn_unpack(int dtype, const void* raw, int ld, float* offset, float* mask, int N, int HW, int n_offset, int n_mask,
                  hipStream_t stream) {
  MR_CHECK_ARG(raw && offset && mask && N > 0 && HW > 0 && n_offset > 0 && n_mask > 0 && ld >= n_offset + n_mask,
               "mr_dcn_unpack: bad arguments");
  const long long items = (long long)N * HW;
  const int grid = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
  DISPATCH_T(dtype, hipLaunchKernelGGL((dcn_unpack_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)raw, ld,
                                       offset, mask, N, HW, n_offset, n_mask));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_dcn_pack_grad(int dtype, const float* doffset, const float* dmask, const float* mask, void* graw, int ld, int N,
                     int HW, int n_offset, int n_mask, hipStream_t stream) {
  MR_CHECK_ARG(doffset && dmask && mask && graw && N > 0 && HW > 0 && ld >= n_offset + n_mask,
               "mr_dcn_pack_grad: bad arguments");
  const long long items = (long long)N * HW;
  const int grid = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
  DISPATCH_T(dtype, hipLaunchKernelGGL((dcn_pack_grad_kernel<T>), dim3(grid), dim3(256), 0, stream, doffset, dmask, mask,
                                       (T*)graw, ld, N, HW, n_offset, n_mask));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
