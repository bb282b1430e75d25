// Fused DCNv2 kernels for gfx950: the column matrix of the reference (assets/ops/dcn/src/deform_conv_cuda_kernel.cu:569-632
// modulated im2col, :634-692 col2im, :694-766 col2im_coord; host GEMM loop deform_conv_cuda.cpp:486-679) never exists in HBM.
//
//   forward   y[p, co]        = sum_{tap, c} sample(x)[p, tap, c] * W[co, tap, c]        dcn2_fwd_fused_kernel
//             sample -> registers -> LDS -> MFMA: the A operand of the implicit GEMM is produced by the bilinear blend
//             of four 16-byte NHWC corner vectors (issued one k-step ahead, blended after the MFMAs of the current one)
//   d offset / d mask          dcn2_coord_fused_kernel
//             gcol[p, (tap, c)] = dy[p, :] . W[:, tap, c] is an MFMA tile that stays in the accumulators; the epilogue dots it
//             with the four corner vectors of x, reduces over the channels of the tile (registers -> 2 shuffles) and adds
//             the three per-(pixel, tap) results into the offset / mask gradients
//   d x       dx[q, c]         = sum_{tap} ( sum_{e in L(q,tap)} w_e dy[p_e, :] ) . W[:, tap, c]   dcn2_dx_fused_kernel
//             the scatter of the reference's col2im (36 f32 atomics on every input element) is inverted once per call into
//             a CSR list per (input pixel, tap) -- the pattern does not depend on the channel -- and the input gradient
//             becomes a GATHER-GEMM: rows of dy are gathered with their bilinear * mask weights into the A tile, the MFMA
//             contracts with W.  No floating-point atomics, no gcol.
//   d W, d bias  dW[co, (tap,c)] = sum_p dy[p, co] * sample(x)[p, tap, c]                 dcn2_wgrad_fused_kernel
//             transpose-read TN GEMM whose B tile is sampled on the fly; split over pixels, f32 atomics on dW (9*C*Co).
//
// Layouts: x / y / dy NHWC in T (bf16 or f32); w_n [Co][taps*C], w_t [taps*C][Co] in T; offsets / mask f32, flat per sample.
// Preconditions of the fused path (host: dcn_fused_ok): C % 64 == 0, Co % 64 == 0, kh*kw <= 9, H, W < 32768.  Other shapes
// take the general kernels of dcn.hip.  MFMA tile code (LDS image, fragment reads) follows igemm_nt_body / igemm_tn_kernel.
#include "dcn_geom.h"
#include "igemm_core.h"
#include "tuning.h"
#include "../../include/megreader_hip.h"

namespace mr {

constexpr int DCN_MAX_TAPS = 9;

template <typename T> __device__ __forceinline__ uint4 pack_vec(const float* v);
template <> __device__ __forceinline__ uint4 pack_vec<float>(const float* v) {
  uint4 o;
  float* po = (float*)&o;
  po[0] = v[0]; po[1] = v[1]; po[2] = v[2]; po[3] = v[3];
  return o;
}
template <> __device__ __forceinline__ uint4 pack_vec<bf16_t>(const float* v) {
  uint4 o;
  bf16_t* po = (bf16_t*)&o;
#pragma unroll
  for (int j = 0; j < 8; ++j) po[j] = (bf16_t)v[j];
  return o;
}

// 64 x BN output tile, 4 waves (2 x 2), BK = 8 vectors of 16 bytes; LDS image [8 k-chunks][ROWS][16 B], slot(row, kc) =
// kc*ROWS + (row ^ kc) (conflict-free ds_write_b128 / ds_read_b128, see igemm_core.h)
template <typename T, int BN>
struct DcnNt {
  static constexpr int BM = 64;
  static constexpr int VEC = VecOf<T>::N, BK = 8 * VEC, AI = BM / 32, BI = BN / 32;
  static constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  typedef typename Mma<T>::Frag Frag;

  static __device__ __forceinline__ void store(uint4* sA, uint4* sB, const uint4* ra, const uint4* rb, int kc, int r0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) sA[kc * BM + ((r0 + 32 * i) ^ kc)] = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) sB[kc * BN + ((r0 + 32 * i) ^ kc)] = rb[i];
  }
  static __device__ __forceinline__ void mma(const uint4* sA, const uint4* sB, f32x4 (&acc)[TN][TM], int wm_, int wn_,
                                             int l15, int lg) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kcr = ks * 4 + lg;
      Frag fa[TM], fb[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[j] = *(const Frag*)&sA[kcr * BM + ((wm_ * WTM + j * 16 + l15) ^ kcr)];
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[i] = *(const Frag*)&sB[kcr * BN + ((wn_ * WTN + i * 16 + l15) ^ kcr)];
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::run(acc[i][j], fb[i], fa[j]);   // D[n][m]: lane = 4 consecutive n of one m
    }
  }
};

struct DcnFusedArgs {
  const void* x;
  const void* w;       // fwd: w_n [Co][taps*C]; coord / dx: w_t [taps*C][Co]
  const void* dy;
  const float* bias;
  const float* offset;
  const float* mask;
  void* y;
  float* doffset;
  float* dmask;
  float* dx;
  void* dx_t;            // dx in the compute dtype, OVERWRITTEN (no zero fill, no conversion pass); only with tsplit == 1
  const int* start;      // CSR row starts, [Q*taps + 1]
  const int2* entries;   // CSR entries {output pixel p, bits of the bilinear * mask weight}
  float* y32;            // fwd with tap splits: f32 [tsplit][P][Co] slabs of partial sums, summed in order by dcn_finish_kernel
  DcnGeom g;
  int Co, P, Q;
  int tsplit;            // tap groups (grid.y): the k-loop of a workgroup covers taps / tsplit taps; partial results are
                         // combined with f32 atomics.  Small layers (batch 2 of the DB detector: 52 workgroups running
                         // 72 dependent k-steps each) are latency chains without it.
};

// ------------------------------------------------------------------------------------------------ forward
// 32-bit addressing through buffer resources: a corner is one byte offset (or DCN_OOB, which the hardware answers with zeros:
// no branch, no 64-bit address arithmetic), computed ONCE per (row, tap) into an LDS table next to the four blend weights.
// Per k-step a thread then spends one LDS read + one add per corner on addressing (the first version recomputed hl/wl, the
// inside tests and a 64-bit multiply-add per corner per k-step: 292 VALU instructions per 16 MFMAs, SQ counters in
// profiles/r03_pmc_sq_dcn_fused_pass{1,2}.txt).
constexpr unsigned DCN_OOB = 0x80000000u;   // >= num_records of make_rsrc (2 GiB), also after adding a channel offset

__device__ __forceinline__ uint4 buf_ld16(rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

// byte offsets of the four corners of a sample at channel 0 (DCN_OOB outside the image) and the blend weights (mask folded in)
template <typename T>
__device__ __forceinline__ void dcn_corner_table(const DcnGeom& g, int pixbase, const DcnDesc& d, uint4& off, float4& wgt) {
  unsigned o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int h = d.hl + (k >> 1), w = d.wl + (k & 1);
    o[k] = dcn_inside(g, h, w) ? (unsigned)((pixbase + h * g.W + w) * g.C) * (unsigned)sizeof(T) : DCN_OOB;
  }
  off = make_uint4(o[0], o[1], o[2], o[3]);
  wgt = make_float4((1.f - d.lh) * (1.f - d.lw) * d.m, (1.f - d.lh) * d.lw * d.m, d.lh * (1.f - d.lw) * d.m,
                    d.lh * d.lw * d.m);
}

template <typename T>
__device__ __forceinline__ uint4 dcn_blend_w(const uint4 (&raw)[4], const float4& w4) {
  constexpr int VEC = VecOf<T>::N;
  const float wgt[4] = {w4.x, w4.y, w4.z, w4.w};
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const T* pv = (const T*)&raw[k];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] += wgt[k] * to_f32(pv[j]);
  }
  return pack_vec<T>(acc);
}

template <typename T, int BN>
__global__ __launch_bounds__(256) void dcn2_fwd_fused_kernel(DcnFusedArgs a) {
  typedef DcnNt<T, BN> Nt;
  constexpr int BM = Nt::BM, VEC = Nt::VEC, BK = Nt::BK, AI = Nt::AI, BI = Nt::BI, TM = Nt::TM, TN = Nt::TN;
  constexpr unsigned ES = sizeof(T);
  __shared__ uint4 smem[8 * (BM + BN)];
  __shared__ float4 swgt[DCN_MAX_TAPS * BM];
  __shared__ uint4 soff[DCN_MAX_TAPS * BM];
  uint4* sA = smem;
  uint4* sB = smem + 8 * BM;
  const DcnGeom& g = a.g;
  const int taps = g.kh * g.kw, K = taps * g.C;
  const int tid = threadIdx.x;
  const int tiles_n = (a.Co + BN - 1) / BN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kc = tid & 7, r0 = tid >> 3;
  const rsrc_t rX = make_rsrc(a.x), rW = make_rsrc(a.w);

  for (int idx = tid; idx < BM * taps; idx += 256) {
    const int row = idx % BM, tap = idx / BM;
    const int p = m0 + row;
    DcnDesc d = dcn_desc_invalid();
    int pb = 0;
    if (p < a.P) {
      const int wo = p % g.Wo, r = p / g.Wo;
      const int ho = r % g.Ho, n = r / g.Ho;
      d = dcn_desc(g, a.offset, a.mask, n, tap, ho, wo);
      pb = n * g.H * g.W;
    }
    uint4 off;
    float4 wgt;
    dcn_corner_table<T>(g, pb, d, off, wgt);
    soff[tap * BM + row] = off;
    swgt[tap * BM + row] = wgt;
  }
  // weight rows of this thread: byte offset of (row n, k = kc*VEC), DCN_OOB beyond Co
  unsigned boff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int n = n0 + r0 + 32 * i;
    boff[i] = n < a.Co ? (unsigned)(n * K + kc * VEC) * ES : DCN_OOB;
  }
  __syncthreads();

  uint4 raw[AI][4], ra[AI], rb[BI];
  int i_tap = 0, i_c = 0;          // (tap, first channel) of the NEXT k-step to issue
  auto issue = [&](int k0) {
    const unsigned cb = (unsigned)(i_c + kc * VEC) * ES;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const uint4 o = soff[i_tap * BM + r0 + 32 * i];
      raw[i][0] = buf_ld16(rX, o.x + cb);
      raw[i][1] = buf_ld16(rX, o.y + cb);
      raw[i][2] = buf_ld16(rX, o.z + cb);
      raw[i][3] = buf_ld16(rX, o.w + cb);
    }
    const unsigned kb = (unsigned)k0 * ES;
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = buf_ld16(rW, boff[i] + kb);
    i_c += BK;
    if (i_c == g.C) { i_c = 0; ++i_tap; }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm_ = wave & 1, wn_ = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tper = taps / a.tsplit;
  const int k_begin = blockIdx.y * tper * g.C, k_end = k_begin + tper * g.C;
  i_tap = blockIdx.y * tper;
  int f_tap = i_tap, f_c = 0;      // (tap, first channel) of the k-step whose loads are in `raw`
  issue(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
    for (int i = 0; i < AI; ++i) ra[i] = dcn_blend_w<T>(raw[i], swgt[f_tap * BM + r0 + 32 * i]);
    f_c += BK;
    if (f_c == g.C) { f_c = 0; ++f_tap; }
    Nt::store(sA, sB, ra, rb, kc, r0);
    __syncthreads();
    if (k0 + BK < k_end) issue(k0 + BK);   // corner / weight loads stay in flight under the MFMAs
    Nt::mma(sA, sB, acc, wm_, wn_, l15, lg);
    __syncthreads();
  }

  if (a.tsplit > 1) {   // this tap group's partial sums go to ITS slab with plain stores; dcn_finish_kernel adds the slabs in tap
                        // order, adds the bias and converts: the same bits every run (round 4 added them with f32 atomics, and
                        // the DB step's forward pass differed from run to run)
    float* __restrict__ slab = a.y32 + (long long)blockIdx.y * a.P * a.Co;
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int m = m0 + wm_ * Nt::WTM + j * 16 + l15;
        const int n = n0 + wn_ * Nt::WTN + i * 16 + lg * 4;
        if (m >= a.P || n >= a.Co) continue;
        *(f32x4*)(slab + (long long)m * a.Co + n) = acc[i][j];
      }
    return;
  }
  T* __restrict__ Y = (T*)a.y;
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int m = m0 + wm_ * Nt::WTM + j * 16 + l15;
      const int n = n0 + wn_ * Nt::WTN + i * 16 + lg * 4;
      if (m >= a.P || n >= a.Co) continue;
      f32x4 v = acc[i][j];
      if (a.bias) {
        const f32x4 b = *(const f32x4*)(a.bias + n);
        v += b;
      }
      store4(Y + (long long)m * a.Co + n, v);
    }
}

// y[m, n] = T(sum over the tap groups' slabs, in tap order, of y32[t][m, n]  + bias[n])   (forward with tap splits)
template <typename T>
__global__ void dcn_finish_kernel(const float* __restrict__ y32, const float* __restrict__ bias, T* __restrict__ y,
                                  long long total4, int Co, int nslab) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = ((const f32x4*)y32)[i];
    for (int t = 1; t < nslab; ++t) v += ((const f32x4*)y32)[t * total4 + i];
    if (bias) {
      const int n = (int)((i * 4) % Co);
      v += *(const f32x4*)(bias + n);
    }
    store4(y + i * 4, v);
  }
}

// tap groups for a launch of `tiles` workgroups (taps = 9): only launches that leave most of the chip idle are split -- the
// partial results meet in f32 slabs + an ordered sum-and-convert pass (forward) or f32 atomics (backward), which cost more than the shorter k-loop saves
// once ~100 workgroups exist (measured at batch 2: layer2.1 forward with 200 tiles 29 us unsplit, 87 us split 3-way).
static int dcn_tap_split(long long tiles, int taps) {
  if (taps != 9) return 1;
  return tiles < 96 ? 9 : 1;
}

// ------------------------------------------------------------------------------------------------ d offset / d mask
// GEMM: gcol[p, n] = sum_co dy[p, co] * w_t[n][co], n = tap*C + c; a BN-column tile lies inside ONE tap (C % BN == 0).
template <typename T, int BN>
__global__ __launch_bounds__(256) void dcn2_coord_fused_kernel(DcnFusedArgs a) {
  typedef DcnNt<T, BN> Nt;
  constexpr int BM = Nt::BM, VEC = Nt::VEC, BK = Nt::BK, AI = Nt::AI, BI = Nt::BI, TM = Nt::TM, TN = Nt::TN;
  __shared__ uint4 smem[8 * (BM + BN)];
  uint4* sA = smem;
  uint4* sB = smem + 8 * BM;
  const DcnGeom& g = a.g;
  const int taps = g.kh * g.kw, NB = taps * g.C, K = a.Co;
  const int tid = threadIdx.x;
  const int tiles_n = NB / BN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kc = tid & 7, r0 = tid >> 3;
  const T* __restrict__ DY = (const T*)a.dy;
  const T* __restrict__ WT = (const T*)a.w;
  const T* __restrict__ X = (const T*)a.x;

  uint4 ra[AI], rb[BI];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int p = m0 + r0 + 32 * i;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (p < a.P) ra[i] = ldg16(DY + (long long)p * K + k0 + kc * VEC);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = ldg16(WT + (long long)(n0 + r0 + 32 * i) * K + k0 + kc * VEC);
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm_ = wave & 1, wn_ = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the epilogue's sample descriptors (three dependent f32 loads per pixel) are fetched BEFORE the GEMM loop, which hides them
  const int tap = n0 / g.C, cbase = n0 - tap * g.C;
  const long long hw = (long long)g.Ho * g.Wo;
  DcnDesc dsc[TM];
  int e_n[TM], e_ho[TM], e_wo[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm_ * Nt::WTM + j * 16 + l15;
    dsc[j] = dcn_desc_invalid();
    e_n[j] = 0; e_ho[j] = 0; e_wo[j] = 0;
    if (m < a.P) {
      e_wo[j] = m % g.Wo;
      const int r = m / g.Wo;
      e_ho[j] = r % g.Ho;
      e_n[j] = r / g.Ho;
      dsc[j] = dcn_desc(g, a.offset, a.mask, e_n[j], tap, e_ho[j], e_wo[j]);
    }
  }

  load(0);
  // optionally the epilogue's corner vectors of x (4 corners x TN x 8 bytes per pixel) are requested now as well and land
  // while the GEMM loop runs
  // (measured: prefetching the x corners for bf16 costs 64 registers -> 3 instead of 6 workgroups per CU: 188 us per layer
  // instead of 144 us; the kernel is latency-bound per tile and lives on occupancy.  Kept as a compile-time switch.)
  constexpr bool PREFETCH_X = false;
  typedef typename std::conditional<sizeof(T) == 2, uint2, uint4>::type XV;   // 4 channels of T
  XV xq[PREFETCH_X ? TM : 1][4][TN];
  auto corner_ptr = [&](int j, int k) -> const T* {
    const int h = dsc[j].hl + (k >> 1), w = dsc[j].wl + (k & 1);
    if (!dcn_inside(g, h, w)) return nullptr;
    return X + ((long long)(e_n[j] * g.H * g.W + h * g.W + w)) * g.C + cbase + wn_ * Nt::WTN + lg * 4;
  };
  if constexpr (PREFETCH_X) {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const T* xp = corner_ptr(j, k);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          XV z;
          __builtin_memset(&z, 0, sizeof(z));
          xq[j][k][i] = xp ? *(const XV*)(xp + i * 16) : z;
        }
      }
  }
  for (int k0 = 0; k0 < K; k0 += BK) {
    Nt::store(sA, sB, ra, rb, kc, r0);
    __syncthreads();
    if (k0 + BK < K) load(k0 + BK);
    Nt::mma(sA, sB, acc, wm_, wn_, l15, lg);
    __syncthreads();
  }

  // epilogue: lane (l15, lg) holds gcol[pixel m][4 channels] per (i, j).  S_k = sum_c gcol * x[corner k]; the three
  // gradients are linear combinations of S_1..S_4 (deform_conv_cuda_kernel.cu:694-766 restated per corner).
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm_ * Nt::WTM + j * 16 + l15;
    const bool live = m < a.P;
    const int n_img = e_n[j], ho = e_ho[j], wo = e_wo[j];
    const DcnDesc d = dsc[j];
    float S[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T* xp = PREFETCH_X ? nullptr : corner_ptr(j, k);
      if (!PREFETCH_X && !xp) continue;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        f32x4 xv;
        if constexpr (PREFETCH_X) {
          const T* pv = (const T*)&xq[j][k][i];
          xv[0] = to_f32(pv[0]); xv[1] = to_f32(pv[1]); xv[2] = to_f32(pv[2]); xv[3] = to_f32(pv[3]);
        } else {
          xv = load4(xp + i * 16);
        }
        const f32x4 gv = acc[i][j];
        S[k] += gv[0] * xv[0] + gv[1] * xv[1] + gv[2] * xv[2] + gv[3] * xv[3];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      S[k] += __shfl_xor(S[k], 16, 64);
      S[k] += __shfl_xor(S[k], 32, 64);
    }
    if (lg == 0 && live && d.hl != -2) {
      const float lh = d.lh, lw = d.lw;
      const float dm = (1.f - lh) * (1.f - lw) * S[0] + (1.f - lh) * lw * S[1] + lh * (1.f - lw) * S[2] + lh * lw * S[3];
      const float dh = d.m * (-(1.f - lw) * S[0] - lw * S[1] + (1.f - lw) * S[2] + lw * S[3]);
      const float dw = d.m * (-(1.f - lh) * S[0] + (1.f - lh) * S[1] - lh * S[2] + lh * S[3]);
      const long long o = (long long)ho * g.Wo + wo;
      atomicAdd(a.doffset + n_img * g.off_bs + (2 * tap) * hw + o, dh);
      atomicAdd(a.doffset + n_img * g.off_bs + (2 * tap + 1) * hw + o, dw);
      atomicAdd(a.dmask + n_img * g.msk_bs + tap * hw + o, dm);
    }
  }
}

// ------------------------------------------------------------------------------------------------ d x (gather-GEMM)
// rows = input pixels q; A[q, tap*Co + co] = sum_{e in L(q,tap)} w_e dy[p_e, co]; B[c][tap*Co + co] = w_t[(tap*C + c)][co]
//
// The CSR keys are tap-major (key = tap*Q + q), so the entry lists of the 64 rows of a tile for one tap form ONE contiguous
// segment of the entry array.  A k-step (tap, 64-wide co chunk) therefore runs a BALANCED gather: the 256 threads load the
// segment's entries and dy vectors in lock step (entry e -> threads 8e..8e+7, 16 bytes each) into an LDS stage, and only then
// does every (row, chunk) thread sum its row's slice of the stage -- LDS latency, not global latency, multiplied by the
// longest list.  (One thread walking its own row's list against global memory measured 445-600 us per layer with 71 % of the
// wave cycles in s_waitcnt / barrier: the k-step of a workgroup costs max-over-64-rows dependent load pairs.)
// Pipeline: the entries of step s+2 and the dy vectors of step s+1 are in flight while step s runs its MFMAs.
constexpr int DX_LCAP = 256;   // entries staged per round (32 KB of LDS at 128 B per entry)

template <typename T, int BN>
__global__ __launch_bounds__(256) void dcn2_dx_fused_kernel(DcnFusedArgs a) {
  typedef DcnNt<T, BN> Nt;
  constexpr int BM = Nt::BM, VEC = Nt::VEC, BK = Nt::BK, AI = Nt::AI, BI = Nt::BI, TM = Nt::TM, TN = Nt::TN;
  constexpr int GJ = DX_LCAP / 32;
  __shared__ uint4 smem[8 * (BM + BN)];
  __shared__ uint4 sG[DX_LCAP * 8];
  __shared__ int2 sE[DX_LCAP];
  __shared__ int sstart[DCN_MAX_TAPS * (BM + 1)];
  uint4* sA = smem;
  uint4* sB = smem + 8 * BM;
  const DcnGeom& g = a.g;
  const int taps = g.kh * g.kw, Co = a.Co, K = taps * Co;
  const int tid = threadIdx.x;
  const int tiles_n = g.C / BN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kc = tid & 7, r0 = tid >> 3;
  const T* __restrict__ DY = (const T*)a.dy;
  const T* __restrict__ WT = (const T*)a.w;

  for (int idx = tid; idx < taps * (BM + 1); idx += 256) {
    const int tap = idx / (BM + 1), r = idx - tap * (BM + 1);
    const int q = min(m0 + r, a.Q);
    sstart[idx] = a.start[(long long)tap * a.Q + q];
  }
  __syncthreads();

  uint4 ra[AI], rb[BI], gq[GJ];
  int2 entC[GJ], entN[GJ];
  auto load_b = [&](int k0) {
    const int tap = k0 / Co;
    const int co = k0 - tap * Co + kc * VEC;
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = ldg16(WT + ((long long)(tap * g.C + n0 + r0 + 32 * i)) * Co + co);
  };
  // entries [base, base + cnt) of the tile's segment for the tap of k-step k0: entry e is handled by threads 8e .. 8e+7
  auto load_entries = [&](int2 (&ent)[GJ], int base, int cnt) {
#pragma unroll
    for (int j = 0; j < GJ; ++j) {
      const int e = r0 + 32 * j;
      ent[j] = make_int2(-1, 0);
      if (e < cnt) ent[j] = a.entries[base + e];
    }
  };
  auto load_dy = [&](const int2 (&ent)[GJ], int k0) {
    const int tap = k0 / Co;
    const int co = k0 - tap * Co + kc * VEC;
#pragma unroll
    for (int j = 0; j < GJ; ++j) {
      gq[j] = make_uint4(0, 0, 0, 0);
      if (ent[j].x >= 0) gq[j] = ldg16(DY + (long long)ent[j].x * Co + co);
    }
  };
  auto stage = [&](const int2 (&ent)[GJ]) {
#pragma unroll
    for (int j = 0; j < GJ; ++j) {
      const int e = r0 + 32 * j;
      sG[e * 8 + kc] = gq[j];
      if (kc == 0) sE[e] = ent[j];
    }
  };
  float accv[AI][VEC];
  auto reduce_rows = [&](int tap, int base, int cnt) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = r0 + 32 * i;
      const int lo = min(max(sstart[tap * (BM + 1) + row] - base, 0), cnt);
      const int hi = min(max(sstart[tap * (BM + 1) + row + 1] - base, 0), cnt);
      for (int idx = lo; idx < hi; ++idx) {
        const float w = __int_as_float(sE[idx].y);
        const uint4 v = sG[idx * 8 + kc];
        const T* pv = (const T*)&v;
#pragma unroll
        for (int j = 0; j < VEC; ++j) accv[i][j] += w * to_f32(pv[j]);
      }
    }
  };
  auto seg_base = [&](int k0) { return sstart[(k0 / Co) * (BM + 1)]; };
  auto seg_cnt = [&](int k0) {
    const int t = k0 / Co;
    return min(sstart[t * (BM + 1) + BM] - sstart[t * (BM + 1)], DX_LCAP);   // first round only
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm_ = wave & 1, wn_ = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tper = taps / a.tsplit;
  const int k_begin = blockIdx.y * tper * Co, k_end = k_begin + tper * Co;
  load_entries(entC, seg_base(k_begin), seg_cnt(k_begin));
  load_b(k_begin);
  load_dy(entC, k_begin);
  if (k_begin + BK < k_end) load_entries(entN, seg_base(k_begin + BK), seg_cnt(k_begin + BK));
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    const int tap = k0 / Co;
    const int s0 = sstart[tap * (BM + 1)], s1 = sstart[tap * (BM + 1) + BM];
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
      for (int j = 0; j < VEC; ++j) accv[i][j] = 0.f;
    // round 0: the stage registers were filled while the previous step's MFMAs ran
    stage(entC);
    __syncthreads();
    reduce_rows(tap, s0, min(s1 - s0, DX_LCAP));
    for (int base = s0 + DX_LCAP; base < s1; base += DX_LCAP) {   // segments longer than the stage (rare): unpipelined rounds
      const int cnt = min(s1 - base, DX_LCAP);
      int2 ent[GJ];
      load_entries(ent, base, cnt);
      load_dy(ent, k0);
      __syncthreads();            // everybody is done reading the previous round's stage
      stage(ent);
      __syncthreads();
      reduce_rows(tap, base, cnt);
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) ra[i] = pack_vec<T>(accv[i]);
    Nt::store(sA, sB, ra, rb, kc, r0);
    __syncthreads();
    if (k0 + BK < k_end) {
#pragma unroll
      for (int j = 0; j < GJ; ++j) entC[j] = entN[j];
      load_dy(entC, k0 + BK);
      load_b(k0 + BK);
      if (k0 + 2 * BK < k_end) load_entries(entN, seg_base(k0 + 2 * BK), seg_cnt(k0 + 2 * BK));
    }
    Nt::mma(sA, sB, acc, wm_, wn_, l15, lg);
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int q = m0 + wm_ * Nt::WTM + j * 16 + l15;
      const int n = n0 + wn_ * Nt::WTN + i * 16 + lg * 4;
      if (q >= a.Q) continue;
      if (a.dx_t) {                      // the complete sum of this element: written once, in the compute dtype
        store4((T*)a.dx_t + (long long)q * g.C + n, acc[i][j]);
        continue;
      }
      float* dst = a.dx + (long long)q * g.C + n;
      if (a.tsplit > 1) {                // several workgroups (tap groups) add into the same elements
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dst + e, acc[i][j][e]);
      } else {
        f32x4 v = *(const f32x4*)dst;    // accumulate semantics (the caller hands a zeroed buffer, like the reference)
        v += acc[i][j];
        *(f32x4*)dst = v;
      }
    }
}

// ------------------------------------------------------------------------------------------------ bf16: materialised gcol
// Round 6.  The fused kernels above keep gcol = dy * W in accumulators and pay for it twice: the coord kernel
// runs the whole GEMM again per column tile just to dot it with four corner vectors, and the gather-GEMM of the input gradient is
// a chain of dependent (entries -> dy rows) round trips per 64-deep k-step (0.02-0.04 of the MFMA peak, 0.1-0.3 TB/s).  For bf16
// the cheaper order is the reference's own (deform_conv_cuda.cpp:611-675): ONE dense GEMM gcol[P, taps*C] = dy[P, Co] * W
// (mr_gemm_nt: the tuned NT kernels, bf16 output = 2 bytes per element of traffic), then two bandwidth-bound passes over it:
//   dcn2_coord_gcol_kernel : per (pixel, tap) the C/8 lanes of a group dot gcol's slice with the four corner vectors of x,
//                            reduce over the group with shuffles and WRITE the three gradients (one writer: no atomics);
//   dcn2_dx_gcol_kernel    : per input pixel the C/8 lanes walk the CSR lists of the nine taps and gather
//                            sum_e w_e * gcol[p_e, tap, :] in f32 -- the reference's col2im inverted (no atomics, same bits every run).
// float32 keeps the fused kernels (their f32 accumulators are what the 2e-4 parity bars against the reference extension need).
template <int L>   // lanes per item = C / 8
__global__ __launch_bounds__(256) void dcn2_coord_gcol_kernel(DcnFusedArgs a, const bf16_t* __restrict__ gcol,
                                                               int* __restrict__ count) {
  const DcnGeom& g = a.g;
  const int taps = g.kh * g.kw, C = L * 8;
  const long long item = (long long)blockIdx.x * (256 / L) + threadIdx.x / L;
  const int sub = threadIdx.x % L;
  const long long items = (long long)a.P * taps;
  const bool live = item < items;
  const long long it = live ? item : 0;
  const int p = (int)(it / taps), tap = (int)(it - (long long)p * taps);
  const int wo = p % g.Wo, r = p / g.Wo;
  const int ho = r % g.Ho, n = r / g.Ho;
  const DcnDesc d = dcn_desc(g, a.offset, a.mask, n, tap, ho, wo);
  const bf16_t* __restrict__ X = (const bf16_t*)a.x;
  float S[4] = {0.f, 0.f, 0.f, 0.f};
  if (count != nullptr && live && d.hl != -2 && sub < 4) {
    // first pass of the CSR build rides here (the descriptor is already in registers): lane k counts corner k, exactly as
    // dcn_csr_kernel<false> does -- one launch less per layer in front of the scan
    const int k = sub;
    const int h = d.hl + (k >> 1), w = d.wl + (k & 1);
    const float wk = ((k >> 1) ? d.lh : 1.f - d.lh) * ((k & 1) ? d.lw : 1.f - d.lw) * d.m;
    if (dcn_inside(g, h, w) && wk != 0.f)
      atomicAdd(count + (long long)tap * ((long long)g.N * g.H * g.W) + ((n * g.H + h) * g.W + w), 1);
  }
  if (live && d.hl != -2) {
    const uint4 gv = *(const uint4*)(gcol + it * C + sub * 8);
    const bf16_t* gp = (const bf16_t*)&gv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int h = d.hl + (k >> 1), w = d.wl + (k & 1);
      if (dcn_inside(g, h, w)) {
        const uint4 xv = *(const uint4*)(X + ((long long)(n * g.H + h) * g.W + w) * C + sub * 8);
        const bf16_t* xp = (const bf16_t*)&xv;
#pragma unroll
        for (int e = 0; e < 8; ++e) S[k] += to_f32(gp[e]) * to_f32(xp[e]);
      }
    }
  }
#pragma unroll
  for (int o = 1; o < L; o <<= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) S[k] += __shfl_xor(S[k], o, 64);
  if (sub == 0 && live && d.hl != -2) {
    const float lh = d.lh, lw = d.lw;
    const float dm = (1.f - lh) * (1.f - lw) * S[0] + (1.f - lh) * lw * S[1] + lh * (1.f - lw) * S[2] + lh * lw * S[3];
    const float dh = d.m * (-(1.f - lw) * S[0] - lw * S[1] + (1.f - lw) * S[2] + lw * S[3]);
    const float dw = d.m * (-(1.f - lh) * S[0] + (1.f - lh) * S[1] - lh * S[2] + lh * S[3]);
    const long long hw = (long long)g.Ho * g.Wo, o = (long long)ho * g.Wo + wo;
    a.doffset[n * g.off_bs + (2 * tap) * hw + o] += dh;       // accumulate semantics; this group is the element's only writer
    a.doffset[n * g.off_bs + (2 * tap + 1) * hw + o] += dw;
    a.dmask[n * g.msk_bs + tap * hw + o] += dm;
  }
}

template <int L>
__global__ __launch_bounds__(256) void dcn2_dx_gcol_kernel(DcnFusedArgs a, const bf16_t* __restrict__ gcol) {
  const DcnGeom& g = a.g;
  const int taps = g.kh * g.kw, C = L * 8;
  const int q = blockIdx.x * (256 / L) + threadIdx.x / L;
  const int sub = threadIdx.x % L;
  if (q >= a.Q) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long long row = (long long)taps * C;
  for (int tap = 0; tap < taps; ++tap) {
    const int lo = a.start[(long long)tap * a.Q + q], hi = a.start[(long long)tap * a.Q + q + 1];
    const bf16_t* __restrict__ gt = gcol + (long long)tap * C + sub * 8;
    int e = lo;
    for (; e + 1 < hi; e += 2) {      // two independent (entry -> row) chains in flight
      const int2 e0 = a.entries[e], e1 = a.entries[e + 1];
      const uint4 v0 = *(const uint4*)(gt + (long long)e0.x * row);
      const uint4 v1 = *(const uint4*)(gt + (long long)e1.x * row);
      const float w0 = __int_as_float(e0.y), w1 = __int_as_float(e1.y);
      const bf16_t* p0 = (const bf16_t*)&v0;
      const bf16_t* p1 = (const bf16_t*)&v1;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w0 * to_f32(p0[j]) + w1 * to_f32(p1[j]);
    }
    if (e < hi) {
      const int2 e0 = a.entries[e];
      const uint4 v0 = *(const uint4*)(gt + (long long)e0.x * row);
      const float w0 = __int_as_float(e0.y);
      const bf16_t* p0 = (const bf16_t*)&v0;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w0 * to_f32(p0[j]);
    }
  }
  if (a.dx_t) {
    *(uint4*)((bf16_t*)a.dx_t + (long long)q * C + sub * 8) = pack_vec<bf16_t>(acc);
  } else {
    float* dst = a.dx + (long long)q * C + sub * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] += acc[j];
  }
}

#define g_dcn_gcol MR_TUNE(dcn_gcol)
// the materialised-gcol backward serves: bf16, C a power of two in 64 .. 512 (C / 8 lanes per item fit one wave)
bool dcn_use_gcol(int dtype, int C) {
  return dtype == MR_BF16 && g_dcn_gcol && (C == 64 || C == 128 || C == 256 || C == 512);
}
#define g_dcn_col_fwd MR_TUNE(dcn_col_fwd)
// forward through the column matrix too (bf16, gcol shapes): col is written ONCE by the forward into the caller's col_ws and the
// backward's weight gradient reads it there instead of sampling it again (mr_dcn2_bwd3: col_saved)
bool dcn_use_col_fwd(int dtype, int C) { return dcn_use_gcol(dtype, C) && g_dcn_col_fwd; }

// ------------------------------------------------------------------------------------------------ CSR of the scatter
// key = tap * Q + q (q = input pixel index over the whole batch; tap-major, so that the neighbouring pixels a wave handles
// hit neighbouring counters -- pixel-major keys measured 28 G atomics/s, one cache line per atomic); an entry = (output
// pixel p, bilinear * mask weight).
// Corners outside the image and zero weights (integer sample positions, zero masks) produce no entry.
template <bool FILL>
__global__ __launch_bounds__(256) void dcn_csr_kernel(const float* __restrict__ offset, const float* __restrict__ mask,
                                                      int* __restrict__ count, const int* __restrict__ start,
                                                      int2* __restrict__ entries, DcnGeom g, int P) {
  const int taps = g.kh * g.kw;
  const long long t = blockIdx.x * 256ll + threadIdx.x;
  if (t >= (long long)P * taps) return;
  const int p = (int)(t % P), tap = (int)(t / P);
  const int wo = p % g.Wo, r = p / g.Wo;
  const int ho = r % g.Ho, n = r / g.Ho;
  const DcnDesc d = dcn_desc(g, offset, mask, n, tap, ho, wo);
  if (d.hl == -2) return;
  const float wgt[4] = {(1.f - d.lh) * (1.f - d.lw) * d.m, (1.f - d.lh) * d.lw * d.m, d.lh * (1.f - d.lw) * d.m,
                        d.lh * d.lw * d.m};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int h = d.hl + (k >> 1), w = d.wl + (k & 1);
    if (!dcn_inside(g, h, w) || wgt[k] == 0.f) continue;
    const long long key = (long long)tap * ((long long)g.N * g.H * g.W) + ((n * g.H + h) * g.W + w);   // tap-major
    if (!FILL) {
      atomicAdd(count + key, 1);
    } else {
      const int slot = atomicSub(count + key, 1) - 1;     // count returns to zero: the workspace cleans itself
      entries[start[key] + slot] = make_int2(p, __float_as_int(wgt[k]));
    }
  }
}

// exclusive scan of `n` ints in three launches (2048 items per block)
constexpr int SCAN_ITEMS = 8, SCAN_BLOCK = 256 * SCAN_ITEMS;

__device__ __forceinline__ int block_excl_scan(int v, int* sh, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int s = sh[w];
    if (w < wave) woff += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return woff + incl - v;
}

__global__ __launch_bounds__(256) void scan_sums_kernel(const int* __restrict__ in, int* __restrict__ bsum, int n) {
  __shared__ int sh[4];
  const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j)
    if (base + j < n) s += in[base + j];
  int total;
  block_excl_scan(s, sh, total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scan_blocks_kernel(int* __restrict__ bsum, int nb) {
  __shared__ int sh[4];
  int running = 0;
  for (int c = 0; c < nb; c += 256) {
    const int i = c + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int total;
    const int ex = block_excl_scan(v, sh, total);
    if (i < nb) bsum[i] = running + ex;
    running += total;
  }
}

// bsum = the RAW block totals of scan_sums_kernel: every block adds up the totals in front of it itself (a few hundred ints:
// cheaper than the one-block launch that used to scan them -- 5 us of launch floor per DCN layer at batch 2)
__global__ __launch_bounds__(256) void scan_write_kernel(const int* __restrict__ in, const int* __restrict__ bsum,
                                                         int* __restrict__ out, int n) {
  __shared__ int sh[4];
  __shared__ int sh_off;
  {
    int part = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += bsum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) sh_off = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
  }
  const int block_off = sh_off;
  const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = base + j < n ? in[base + j] : 0;
    s += v[j];
  }
  int total;
  int run = block_off + block_excl_scan(s, sh, total);
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    const int idx = base + j;
    if (idx < n) {
      out[idx] = run;
      run += v[j];
      if (idx == n - 1) out[n] = run;
    }
  }
}

// ------------------------------------------------------------------------------------------------ d W (+ d bias)
// C[co, nb] += sum_{p in split} dy[p, co] * sample(x)[p, nb],  nb = tap*C + c.  128 x 128 tile, 4 waves of 64 x 64,
// LDS image / ds_read_b64_tr_b16 fragment reads of igemm_tn_kernel; the B vectors are blended from four corner loads.
struct DcnWgradArgs {
  const void* dy;
  const void* x;
  const float* offset;
  const float* mask;
  float* dw;       // [Co][taps*C] f32, accumulated
  float* dbias;    // nullable [Co], accumulated by the tile_b == 0 workgroups
  DcnGeom g;
  int Co, P, p_chunk;
};

template <typename T>
__global__ __launch_bounds__(256) void dcn2_wgrad_fused_kernel(DcnWgradArgs a) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int BP = TnCfg<T>::BP;
  constexpr int ROW_VECS = TnCfg<T>::ROW_VECS;
  constexpr int ROW_BYTES = TnCfg<T>::ROW_BYTES;
  constexpr int ROWS_PER_PASS = 256 / ROW_VECS;
  constexpr int NI = BP / ROWS_PER_PASS;
  constexpr bool IS_BF16 = (sizeof(T) == 2);

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BP * ROW_BYTES];
  unsigned char* sA = smem;
  unsigned char* sB = smem + BP * ROW_BYTES;
  const DcnGeom& g = a.g;
  const int taps = g.kh * g.kw, NB = taps * g.C, NA = a.Co;
  const int tid = threadIdx.x;
  const int tiles_b = NB / 128 + (NB % 128 != 0);
  const int tile_b = blockIdx.x % tiles_b, tile_a = blockIdx.x / tiles_b;
  const int na0 = tile_a * 128, nb0 = tile_b * 128;
  const int p_begin = blockIdx.z * a.p_chunk;
  const int p_end = min(a.P, p_begin + a.p_chunk);
  if (p_begin >= p_end) return;
  const T* __restrict__ A = (const T*)a.dy;
  const T* __restrict__ X = (const T*)a.x;

  const int cc = tid % ROW_VECS, rr = tid / ROW_VECS;
  const int ca = na0 + cc * VEC, cb = nb0 + cc * VEC;
  const bool ca_ok = ca < NA, cb_ok = cb < NB;
  const int tap = cb_ok ? cb / g.C : 0;
  const int tc = cb - tap * g.C;

  auto lds_off = [&](int p) -> int {
    if (IS_BF16) {
      const int cp = cc >> 1;
      return p * ROW_BYTES + ((cp ^ tn_hash(p)) << 5) + (cc & 1) * 16;
    } else {
      return p * ROW_BYTES + cc * 16;
    }
  };

  uint4 ra[NI], rb[NI], raw[NI][4];
  DcnDesc ds[NI];
  float4 wq[NI];
  const rsrc_t rX = make_rsrc(a.x);
  const bool do_colsum = a.dbias != nullptr && tile_b == 0;
  typedef typename std::conditional<IS_BF16, float, double>::type CsT;
  CsT csum[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) csum[j] = 0;
  // three-stage software pipeline over the p-steps (one stage measured 10 us per p-step: the corner loads wait for the
  // offset / mask loads they depend on, and the MFMAs queue behind both):
  //   stage_a(s + 2): offset (dh, dw) and mask floats of the rows of step s + 2
  //   stage_b(s + 1): descriptors from the floats fetched a step ago, then the dy vector + four corner loads
  //   step s        : blend -> LDS -> MFMAs
  int q_n[NI], q_h[NI], q_w[NI];        // pixel of stage_a's next call
  int b_n[NI], b_h[NI], b_w[NI];        // pixel + raw floats handed from stage_a to stage_b
  float b_oh[NI], b_ow[NI], b_m[NI];
  bool b_ok[NI];
  int pa = p_begin;                     // first row of stage_a's next call
  int pbq = p_begin;                    // first row of stage_b's next call
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = p_begin + rr + ROWS_PER_PASS * i;
    q_w[i] = p % g.Wo;
    const int t = p / g.Wo;
    q_h[i] = t % g.Ho;
    q_n[i] = t / g.Ho;
  }
  const long long hwo = (long long)g.Ho * g.Wo;
  auto stage_a = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = pa + rr + ROWS_PER_PASS * i;
      b_n[i] = q_n[i]; b_h[i] = q_h[i]; b_w[i] = q_w[i];
      b_ok[i] = p < p_end && cb_ok;
      b_oh[i] = 0.f; b_ow[i] = 0.f; b_m[i] = 0.f;
      if (b_ok[i]) {
        const long long o = (long long)q_h[i] * g.Wo + q_w[i];
        const float* ob = a.offset + q_n[i] * g.off_bs + (2 * tap) * hwo + o;
        b_oh[i] = ob[0];
        b_ow[i] = ob[hwo];
        b_m[i] = a.mask[q_n[i] * g.msk_bs + tap * hwo + o];
      }
      q_w[i] += BP;
      while (q_w[i] >= g.Wo) {
        q_w[i] -= g.Wo;
        if (++q_h[i] == g.Ho) { q_h[i] = 0; ++q_n[i]; }
      }
    }
    pa += BP;
  };
  auto stage_b = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = pbq + rr + ROWS_PER_PASS * i;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (p < p_end && ca_ok) ra[i] = ldg16(A + (long long)p * NA + ca);
      ds[i] = b_ok[i] ? dcn_desc_from(g, tap, b_h[i], b_w[i], b_oh[i], b_ow[i], b_m[i]) : dcn_desc_invalid();
      uint4 off;
      dcn_corner_table<T>(g, b_n[i] * g.H * g.W, ds[i], off, wq[i]);     // 32-bit offsets, DCN_OOB outside: no branches
      const unsigned cb = (unsigned)tc * (unsigned)sizeof(T);
      raw[i][0] = buf_ld16(rX, off.x + cb);
      raw[i][1] = buf_ld16(rX, off.y + cb);
      raw[i][2] = buf_ld16(rX, off.z + cb);
      raw[i][3] = buf_ld16(rX, off.w + cb);
    }
    pbq += BP;
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wa = wave & 1, wb = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage_a();
  stage_b();
  stage_a();
  for (int p0 = p_begin; p0 < p_end; p0 += BP) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int pl = rr + ROWS_PER_PASS * i;
      rb[i] = dcn_blend_w<T>(raw[i], wq[i]);
      *(uint4*)(sA + lds_off(pl)) = ra[i];
      *(uint4*)(sB + lds_off(pl)) = rb[i];
      if (do_colsum) {
        const T* pv = (const T*)&ra[i];
#pragma unroll
        for (int j = 0; j < VEC; ++j) csum[j] += to_f32(pv[j]);
      }
    }
    __syncthreads();
    if (p0 + BP < p_end) {
      stage_b();     // rows of step s + 1: their offset / mask floats were requested a step ago
      stage_a();     // floats of step s + 2
    }

    if constexpr (IS_BF16) {
#pragma unroll
      for (int kk = 0; kk < BP / 32; ++kk) {
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int cpa = (wa * 64 + t * 16) >> 4;
          const int cpb = (wb * 64 + t * 16) >> 4;
          s16x4 x[2], y[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int p = kk * 32 + lg * 8 + hh * 4 + (l15 >> 2);
            const int h = tn_hash(p);
            const int oa = p * ROW_BYTES + ((cpa ^ h) << 5) + (l15 & 3) * 8;
            const int ob = p * ROW_BYTES + ((cpb ^ h) << 5) + (l15 & 3) * 8;
            x[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sA + oa));
            y[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sB + ob));
          }
          union { s16x4 h[2]; bf16x8 v; } ua, ub;
          ua.h[0] = x[0]; ua.h[1] = x[1];
          ub.h[0] = y[0]; ub.h[1] = y[1];
          fa[t] = ua.v;
          fb[t] = ub.v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BP / 4; ++ks) {
        const int p = ks * 4 + lg;
        float fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fa[t] = *(const float*)(sA + p * ROW_BYTES + (wa * 64 + t * 16 + l15) * 4);
          fb[t] = *(const float*)(sB + p * ROW_BYTES + (wb * 64 + t * 16 + l15) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  if (do_colsum) {
    CsT* red = (CsT*)smem;
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[rr * 128 + cc * VEC + j] = csum[j];
    __syncthreads();
    if (tid < 128 && na0 + tid < NA) {
      CsT sum = 0;
      for (int r = 0; r < ROWS_PER_PASS; ++r) sum += red[r * 128 + tid];
      atomicAdd(a.dbias + na0 + tid, (float)sum);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = nb0 + wb * 64 + j * 16 + l15;
      if (col >= NB) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = na0 + wa * 64 + i * 16 + lg * 4 + q;
        if (row >= NA) continue;
        atomicAdd(a.dw + (long long)row * NB + col, acc[i][j][q]);
      }
    }
}

#define g_dcn_fused MR_TUNE(dcn_fused)   // mr_tuning.dcn_fused: 0 forces the general kernels of dcn.hip (A/B and tests)

}  // namespace mr

using namespace mr;

namespace mr {

bool dcn_fused_ok(int dtype, int H, int W, int C, int Co, int kh, int kw) {
  (void)dtype;
  return g_dcn_fused && C % 64 == 0 && Co % 64 == 0 && kh * kw <= DCN_MAX_TAPS && H < 32768 && W < 32768;
}

// the kernels address x / dy / the weight images with 32-bit byte offsets through 2 GiB buffer resources
static bool dcn_fits_32bit(int dtype, long long x_elems, long long dy_elems, long long w_elems) {
  const long long es = dtype == MR_F32 ? 4 : 2;
  return x_elems * es < (1ll << 31) && dy_elems * es < (1ll << 31) && w_elems * es < (1ll << 31);
}

static inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// workspace of the fused backward: [count: Q*taps ints][start: Q*taps + 1 ints][block sums][entries: 4*P*taps int2]
struct DcnWs {
  int* count;
  int* start;
  int* bsum;
  int2* entries;
  void* gcol;        // bf16 [P][taps * C] (materialised-gcol path), else null
  size_t bytes;
  int nkeys, nblocks;
};
static DcnWs dcn_ws_layout(void* base, long long Q, long long P, int taps, long long gcol_bytes = 0) {
  DcnWs w;
  const long long nkeys = Q * taps;
  w.nkeys = (int)nkeys;
  w.nblocks = (int)((nkeys + SCAN_BLOCK - 1) / SCAN_BLOCK);
  size_t o = 0;
  unsigned char* b = (unsigned char*)base;
  w.count = (int*)(b + o); o += align16((size_t)nkeys * 4);
  w.start = (int*)(b + o); o += align16((size_t)(nkeys + 1) * 4);
  w.bsum = (int*)(b + o); o += align16((size_t)w.nblocks * 4);
  w.entries = (int2*)(b + o); o += align16((size_t)P * taps * 4 * 8);
  w.gcol = nullptr;
  if (gcol_bytes > 0) {
    o = (o + 255) & ~(size_t)255;
    w.gcol = b + o;
    o += align16((size_t)gcol_bytes);
  }
  w.bytes = o;
  return w;
}

#define DISPATCH_T(dtype, ...)                                   \
  if ((dtype) == MR_F32) { typedef float T; __VA_ARGS__; }       \
  else if ((dtype) == MR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { mr::set_error("bad dtype %d", (dtype)); return MR_ERR_DTYPE; }

int dcn_fused_fwd(int dtype, const void* x, const void* w_n, const float* bias, const float* offset, const float* mask,
                  void* y, void* ws, const DcnGeom& g, int Co, hipStream_t stream) {
  DcnFusedArgs a = {};
  a.x = x; a.w = w_n; a.bias = bias; a.offset = offset; a.mask = mask; a.y = y; a.g = g; a.Co = Co;
  a.P = g.N * g.Ho * g.Wo;
  MR_CHECK_ARG(dcn_fits_32bit(dtype, (long long)g.N * g.H * g.W * g.C, (long long)a.P * Co, (long long)Co * g.kh * g.kw * g.C),
               "dcn (fused path): x / y / weights must each be smaller than 2 GiB (mr_tuning.dcn_fused = 0 selects the general "
               "kernels)");
  if (dcn_use_col_fwd(dtype, g.C)) {
    // col[P, taps*C] = sampled x (one bandwidth-bound pass), y = col * W on the tuned NT GEMM; col stays in the caller's buffer
    // for the backward (the reference keeps `columns` only as scratch and samples again, deform_conv_cuda.cpp:641-658)
    MR_CHECK_ARG(ws != nullptr, "dcn forward: column workspace missing (mr_dcn2_ws_bytes)");
    const int K = g.kh * g.kw * g.C;
    int rc;
    {
      PhaseScope ph(MR_PH_DCN_IM2COL, 2.0 * g.N * g.H * g.W * g.C + 2.0 * a.P * K + 12.0 * a.P * g.kh * g.kw, stream);
      rc = mr_dcn2_im2col(dtype, x, offset, g.off_bs, mask, g.msk_bs, ws, g.N, g.H, g.W, g.C, g.kh, g.kw, g.stride, g.pad, g.dil,
                          g.Ho, g.Wo, stream);
    }
    if (rc) return rc;
    PhaseScope ph(MR_PH_DCN_FWD, 2.0 * a.P * Co * K, stream);
    return mr_gemm_nt(dtype, ws, K, w_n, K, y, Co, bias, 0, a.P, Co, K, stream);
  }
  const int tiles_m = cdiv(a.P, 64);
  const int bn = Co % 128 == 0 ? 128 : 64;
  PhaseScope ph_fwd(MR_PH_DCN_FWD, 2.0 * a.P * Co * g.kh * g.kw * g.C, stream);
  a.tsplit = dcn_tap_split((long long)tiles_m * (Co / bn), g.kh * g.kw);
  if (a.tsplit > 1) {
    MR_CHECK_ARG(ws != nullptr, "dcn forward: workspace missing (mr_dcn2_ws_bytes)");
    a.y32 = (float*)ws;      // [tsplit][P][Co] f32 slabs, every element written by its tap group (no zero fill)
  }
  if (bn == 128) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_fwd_fused_kernel<T, 128>), dim3(tiles_m * (Co / 128), a.tsplit), dim3(256), 0,
                                         stream, a));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_fwd_fused_kernel<T, 64>), dim3(tiles_m * (Co / 64), a.tsplit), dim3(256), 0,
                                         stream, a));
  }
  MR_CHECK_LAUNCH();
  if (a.tsplit > 1) {
    const long long total4 = (long long)a.P * Co / 4;
    long long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    DISPATCH_T(dtype, hipLaunchKernelGGL((dcn_finish_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream,
                                         (const float*)ws, bias, (T*)y, total4, Co, a.tsplit));
    MR_CHECK_LAUNCH();
  }
  return MR_OK;
}

// forward workspace of the fused path: the f32 accumulator of the tap-split launch (small layers), else nothing
long long dcn_fused_fwd_ws_bytes(int dtype, int C, int N, int Ho, int Wo, int Co, int taps) {
  const long long P = (long long)N * Ho * Wo;
  if (dcn_use_col_fwd(dtype, C)) return P * taps * C * 2;
  const int bn = Co % 128 == 0 ? 128 : 64;
  const int tsplit = dcn_tap_split(((P + 63) / 64) * (Co / bn), taps);
  return tsplit > 1 ? P * Co * 4 * tsplit : 0;
}

static long long dcn_gcol_bytes(int dtype, int C, long long P, int taps) { return dcn_use_gcol(dtype, C) ? P * taps * C * 2 : 0; }
long long dcn_fused_ws_bytes(int dtype, int N, int H, int W, int C, int Ho, int Wo, int taps) {
  const long long P = (long long)N * Ho * Wo;
  return (long long)dcn_ws_layout(nullptr, (long long)N * H * W, P, taps, dcn_gcol_bytes(dtype, C, P, taps)).bytes;
}

// 1 when the fused input-gradient kernel runs un-split for this shape, i.e. can write dx directly in the compute dtype
bool dcn_fused_dx_direct(int dtype, int N, int H, int W, int C, int taps) {
  if (dcn_use_gcol(dtype, C)) return true;     // the gather pass owns every element of dx
  const long long tiles_q = cdivll((long long)N * H * W, 64);
  return dcn_tap_split(tiles_q * (C / (C % 128 == 0 ? 128 : 64)), taps) == 1;
}

// dx_t (nullable, instead of dx32): dx in the compute dtype, overwritten.  flags bit 0: the counter region of `ws` is already
// zero (a workspace that only this function has used since it was zeroed: the fill pass returns every counter to zero).
int dcn_fused_bwd(int dtype, const void* dy, const void* x, const void* w_t, const float* offset, const float* mask,
                  void* ws, float* dx32, void* dx_t, int flags, float* doffset, float* dmask, float* dw, float* dbias,
                  const DcnGeom& g, int Co, const void* col_saved, hipStream_t stream) {
  const int taps = g.kh * g.kw;
  const long long Q = (long long)g.N * g.H * g.W, P = (long long)g.N * g.Ho * g.Wo;
  MR_CHECK_ARG(Q * taps < (1ll << 31) - SCAN_BLOCK && P * taps * 4 < (1ll << 31) && Q * g.C < (1ll << 31),
               "dcn backward: tensor too large for 32-bit indices");
  DcnFusedArgs a = {};
  a.x = x; a.w = w_t; a.dy = dy; a.offset = offset; a.mask = mask; a.doffset = doffset; a.dmask = dmask; a.dx = dx32;
  a.g = g; a.Co = Co; a.P = (int)P; a.Q = (int)Q; a.tsplit = 1;
  a.dx_t = nullptr;
  MR_CHECK_ARG(!(dx32 && dx_t), "dcn backward: dx32 and dx_t are alternatives");
  MR_CHECK_ARG(!dx_t || dcn_fused_dx_direct(dtype, g.N, g.H, g.W, g.C, taps),
               "dcn backward: this shape splits the input gradient over tap groups (ask mr_dcn2_dx_direct first)");
  const int tiles_p = cdiv((int)P, 64);
  // ---- bf16: gcol = dy * W once (tuned NT GEMM), both consumers read it (see "materialised gcol" above)
  const bool use_gcol = dcn_use_gcol(dtype, g.C) && ((doffset && dmask) || dx32 || dx_t);
  const bf16_t* gcol = nullptr;
  if (use_gcol) {
    MR_CHECK_ARG(ws != nullptr, "dcn backward: workspace missing (mr_dcn2_ws_bytes)");
    MR_CHECK_ARG(P * taps * g.C < (1ll << 31), "dcn backward: gcol too large for 32-bit element offsets");
    DcnWs w = dcn_ws_layout(ws, Q, P, taps, dcn_gcol_bytes(dtype, g.C, P, taps));
    gcol = (const bf16_t*)w.gcol;
    PhaseScope ph(MR_PH_DCN_GCOL_GEMM, 2.0 * P * Co * taps * g.C, stream);
    const int rc = mr_gemm_nt(MR_BF16, dy, Co, w_t, Co, w.gcol, taps * g.C, nullptr, 0, (int)P, taps * g.C, Co, stream);
    if (rc) return rc;
  }
#define MR_DCN_GCOL_LAUNCH(KERN, ITEMS, ...)                                                                              \
  {                                                                                                                      \
    const int L_ = g.C / 8;                                                                                              \
    const unsigned nb_ = (unsigned)cdivll((long long)(ITEMS), 256 / L_);                                                 \
    if (L_ == 8) hipLaunchKernelGGL((KERN<8>), dim3(nb_), dim3(256), 0, stream, a, gcol, ##__VA_ARGS__);                 \
    else if (L_ == 16) hipLaunchKernelGGL((KERN<16>), dim3(nb_), dim3(256), 0, stream, a, gcol, ##__VA_ARGS__);          \
    else if (L_ == 32) hipLaunchKernelGGL((KERN<32>), dim3(nb_), dim3(256), 0, stream, a, gcol, ##__VA_ARGS__);          \
    else hipLaunchKernelGGL((KERN<64>), dim3(nb_), dim3(256), 0, stream, a, gcol, ##__VA_ARGS__);                        \
    MR_CHECK_LAUNCH();                                                                                                   \
  }
  // The three parts below are independent of each other (each reads dy / x / w / offset / mask and writes its own outputs):
  // a caller may ask for any subset by passing null for the outputs of the others, e.g. to run them on parallel streams.
  // ---- offset / mask gradients (gcol tiles stay in registers)
  const double es = dtype == MR_F32 ? 4.0 : 2.0;
  const double gemm_flops = 2.0 * P * Co * taps * g.C;
  bool counted = false;      // the CSR's counting pass already ran (inside the coordinate pass)
  if (doffset && dmask && use_gcol) {
    // algorithmic bytes: gcol + x once, offsets / mask read, their three gradients read-modify-written
    PhaseScope ph(MR_PH_DCN_COORD, 2.0 * P * taps * g.C + es * Q * g.C + 9.0 * 4 * 3 * P * taps, stream);
    int* count_here = nullptr;
    if (dx32 || dx_t) {      // the CSR build follows: its counting pass rides in this launch (the counters must be zero NOW)
      DcnWs w = dcn_ws_layout(ws, Q, P, taps, dcn_gcol_bytes(dtype, g.C, P, taps));
      if (!(flags & 1) && hipMemsetAsync(w.count, 0, (size_t)w.nkeys * 4, stream) != hipSuccess) {
        mr::set_error("dcn backward: hipMemsetAsync failed");
        return MR_ERR_LAUNCH;
      }
      count_here = w.count;
      counted = true;
    }
    MR_DCN_GCOL_LAUNCH(dcn2_coord_gcol_kernel, P * taps, count_here)
  } else if (doffset && dmask) {
    PhaseScope ph(MR_PH_DCN_COORD, gemm_flops, stream);
    if (g.C % 128 == 0) {
      DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_coord_fused_kernel<T, 128>), dim3(tiles_p * (taps * g.C / 128)), dim3(256), 0,
                                           stream, a));
    } else {
      DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_coord_fused_kernel<T, 64>), dim3(tiles_p * (taps * g.C / 64)), dim3(256), 0,
                                           stream, a));
    }
    MR_CHECK_LAUNCH();
  }
  // ---- input gradient: CSR of the scatter pattern, then the gather-GEMM
  if (dx32 || dx_t) {
    MR_CHECK_ARG(ws != nullptr, "dcn backward: workspace missing (mr_dcn2_ws_bytes)");
    DcnWs w = dcn_ws_layout(ws, Q, P, taps, dcn_gcol_bytes(dtype, g.C, P, taps));
    if (!counted && !(flags & 1) && hipMemsetAsync(w.count, 0, (size_t)w.nkeys * 4, stream) != hipSuccess) {
      mr::set_error("dcn backward: hipMemsetAsync failed");
      return MR_ERR_LAUNCH;
    }
    const unsigned items = (unsigned)cdivll(P * taps, 256);
    {
    // bytes: offsets + mask twice, counters (atomics) twice, the scan over Q * taps keys, the entries
    PhaseScope ph(MR_PH_DCN_CSR, 2.0 * 12 * P * taps + 5.0 * 4 * Q * taps + 8.0 * 4 * P * taps, stream);
    if (!counted)
      hipLaunchKernelGGL((dcn_csr_kernel<false>), dim3(items), dim3(256), 0, stream, offset, mask, w.count,
                         (const int*)nullptr, (int2*)nullptr, g, (int)P);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(w.nblocks), dim3(256), 0, stream, (const int*)w.count, w.bsum, w.nkeys);
    hipLaunchKernelGGL(scan_write_kernel, dim3(w.nblocks), dim3(256), 0, stream, (const int*)w.count, (const int*)w.bsum,
                       w.start, w.nkeys);
    hipLaunchKernelGGL((dcn_csr_kernel<true>), dim3(items), dim3(256), 0, stream, offset, mask, w.count,
                       (const int*)w.start, w.entries, g, (int)P);
    }
    MR_CHECK_LAUNCH();
    a.start = w.start;
    a.entries = w.entries;
    a.dx_t = dx_t;
    if (use_gcol) {
      // bytes: every gcol element once, the CSR (entries + row starts), dx written once
      PhaseScope ph(MR_PH_DCN_DX, 2.0 * P * taps * g.C + 8.0 * 4 * P * taps + 4.0 * Q * taps + (dx_t ? es : 8.0) * Q * g.C, stream);
      MR_DCN_GCOL_LAUNCH(dcn2_dx_gcol_kernel, Q)
    } else {
    PhaseScope ph(MR_PH_DCN_DX, gemm_flops, stream);
    const int tiles_q = cdiv((int)Q, 64);
    a.tsplit = dcn_tap_split((long long)tiles_q * (g.C / (g.C % 128 == 0 ? 128 : 64)), taps);
    if (g.C % 128 == 0) {
      DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_dx_fused_kernel<T, 128>), dim3(tiles_q * (g.C / 128), a.tsplit), dim3(256), 0,
                                           stream, a));
    } else {
      DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_dx_fused_kernel<T, 64>), dim3(tiles_q * (g.C / 64), a.tsplit), dim3(256), 0,
                                           stream, a));
    }
    a.tsplit = 1;
    MR_CHECK_LAUNCH();
    }
  }
#undef MR_DCN_GCOL_LAUNCH
  // ---- weight / bias gradients
  if (dw && dcn_use_gcol(dtype, g.C) && ws != nullptr) {
    // bf16: the sampled column matrix col[P, taps*C] is written ONCE into the buffer gcol just left (the reference re-uses
    // `columns` the same way, deform_conv_cuda.cpp:641-658) and dW = dy^T * col runs on the tuned TN GEMM kernel -- the fused
    // kernel below samples every B tile again per 128-row output tile and reduces its pixel splits with f32 atomics
    // (237 us per layer at batch 16 against ~110 us for the two launches here).
    DcnWs w = dcn_ws_layout(ws, Q, P, taps, dcn_gcol_bytes(dtype, g.C, P, taps));
    if (col_saved != nullptr) {          // the forward's column matrix is still there (mr_dcn2_bwd3)
      PhaseScope ph(MR_PH_DCN_WGRAD, gemm_flops, stream);
      return mr_gemm_tn(dtype, dy, Co, col_saved, taps * g.C, dw, taps * g.C, (int)P, Co, taps * g.C, 0, dbias, stream);
    }
    int rc;
    {
      PhaseScope ph(MR_PH_DCN_IM2COL, es * Q * g.C + 2.0 * P * taps * g.C + 12.0 * P * taps, stream);
      rc = mr_dcn2_im2col(dtype, x, offset, g.off_bs, mask, g.msk_bs, w.gcol, g.N, g.H, g.W, g.C, g.kh, g.kw, g.stride, g.pad,
                          g.dil, g.Ho, g.Wo, stream);
    }
    if (rc) return rc;
    PhaseScope ph(MR_PH_DCN_WGRAD, gemm_flops, stream);
    return mr_gemm_tn(dtype, dy, Co, w.gcol, taps * g.C, dw, taps * g.C, (int)P, Co, taps * g.C, 0, dbias, stream);
  } else if (dw) {
    PhaseScope ph(MR_PH_DCN_WGRAD, gemm_flops, stream);
    DcnWgradArgs wa = {};
    wa.dy = dy; wa.x = x; wa.offset = offset; wa.mask = mask; wa.dw = dw; wa.dbias = dbias; wa.g = g; wa.Co = Co;
    wa.P = (int)P;
    const int BP = dtype == MR_BF16 ? 64 : 16;
    const int tiles = cdiv(Co, 128) * cdiv(taps * g.C, 128);
    int splits = cdiv(512, tiles);                         // ~2 workgroups per CU
    const int max_splits = (int)cdivll(P, 2 * BP);         // at least two p-steps per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int p_chunk = cdiv(cdiv((int)P, splits), BP) * BP;
    splits = cdiv((int)P, p_chunk);
    wa.p_chunk = p_chunk;
    DISPATCH_T(dtype, hipLaunchKernelGGL((dcn2_wgrad_fused_kernel<T>), dim3(tiles, 1, splits), dim3(256), 0, stream, wa));
    MR_CHECK_LAUNCH();
  } else if (dbias) {
    return mr_colsum(dtype, dy, dbias, (int)P, Co, Co, 0, stream);
  }
  return MR_OK;
}

}  // namespace mr
