// NT implicit-GEMM kernel v4 ("phased"): 256 x 256 tile, 8 waves (2 x 4, wave tile 128 x 64), BK = 64, bf16.
//
// Same operands, same LDS image and same result as igemm_nt_big_kernel (igemm_core.h) -- what changes is the
// SCHEDULE.  The v3 kernel runs one k-step as "barrier; issue 8 LDS-DMAs; 24 fragment reads + 64 MFMAs; drain
// vmcnt(0); barrier": rocprofv3 shows its waves parked at s_waitcnt / s_barrier for 43 % of their cycles with the MFMA
// pipe 35 % busy (profiles/r02_pmc_sq_pass1_v1.txt).  Here a K-tile is four PHASES of 16 MFMAs (one 64 x 32 quadrant
// of the wave tile x K = 64):
//
//   phase   fragments read (ds_read_b128)        LDS-DMA issued (half-tile = 128 rows x 64 k)   MFMAs
//   q0      A0 (8), B0 (4)  + prefetch B1 (4)     BH0 of K-tile c+1                              A0 x B0
//   q1      prefetch A1 (8)                       BH1 of K-tile c+1                              A0 x B1
//           ---- lgkmcnt(0) ; s_barrier (every wave has retired its A reads of K-tile c) ----
//   q2      re-read B0 (4) for q3                 AH0 of K-tile c+2                              A1 x B1
//   q3      --                                    AH1 of K-tile c+2                              A1 x B0
//           ---- s_waitcnt vmcnt(4) ; s_barrier  (K-tile c+1 has landed; 2 half-tiles stay in flight) ----
//
// * the fragment reads of a quadrant are issued one phase before its MFMAs, so a wave's LDS latency is covered by its
//   own MFMAs; B0 is read twice per K-tile instead of being held (the wave runs at the 256-register limit of two
//   waves per SIMD: 128 accumulators + at most 80 fragment registers live);
// * LDS-DMAs are never drained: each half-tile has 3-6 phases to land, the wait is a COUNTED vmcnt, and the barriers
//   are raw s_barrier (a __syncthreads() would emit vmcnt(0));
// * two barriers per K-tile (64 MFMAs per wave between barriers instead of 32 with a full drain).
// Hazards (LDS-DMA is ordered against ds_read only by the issuing wave's vmcnt + a barrier the reader has passed):
//   RAW  K-tile c+1 is first read in (c+1, q0), after the vmcnt + barrier that closes (c, q3);
//   WAR  AH(c+2) overwrites AH(c): its last reads (A1 of K-tile c, issued in q1) are retired by the lgkmcnt(0) in front
//        of the mid barrier; BH(c+1) overwrites BH(c-1), whose last reads (the B0 re-read in (c-1, q2)) are consumed by
//        the q3 MFMAs, i.e. retired before the barrier closing (c-1, q3).
#pragma once
#include "igemm_core.h"

#ifndef MR_P8_SCHED
#define MR_P8_SCHED() __builtin_amdgcn_sched_barrier(0)
#endif

namespace mr {

// ABL (measurement only, results are wrong for ABL != 0): 1 = no LDS-DMA inside the K loop (stage once), 2 = no
// fragment reads inside the K loop, 3 = both (MFMAs + barriers only) -- the ablation ladder behind DESIGN.md's
// "what bounds the big-tile kernel".
template <int AMODE, typename Epi, int ABL = 0>
__global__ __launch_bounds__(512) void igemm_nt_p8_kernel(NtArgs a, ConvGeom g, Epi epi) {
  typedef bf16_t T;
  typedef Mma<T>::Frag Frag;
  constexpr int VEC = 8, BK = 64, BM = 256, BN = 256;
  constexpr int STAGE_VECS = (BM + BN) * 8;   // 16-byte vectors per stage: A 2048 | B 2048
  static_assert(AMODE == 0 || AMODE == 2, "phased kernel: dense or fast conv gather only");

  extern __shared__ uint4 smem_p8[];
  uint4* smem = smem_p8;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M - a.m_begin + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;   // XCD-aware map, see igemm_nt_big_kernel
  const int tile_m = (slot / tiles_n) * 8 + xcd, tile_n = slot % tiles_n;
  if (tile_m >= tiles_m) return;
  const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;
  const int lrow = lane >> 3, lpc = lane & 7;

  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;
  (void)a.zero;
  const rsrc_t rsA = make_rsrc(A), rsB = make_rsrc(B);

  // ---- staging descriptors.  Half-tile hh (rows hh*128 ..) = 16 groups of 8 rows; this lane stages row lrow of
  // groups hh*16 + wave and hh*16 + wave + 8, 16-byte chunk lpc (source chunk swizzled, LDS image lane-linear).
  auto kc_of = [&](int gi) { return (lpc ^ (((gi * 8 + lrow) >> 1) & 7)) * VEC; };
  int a_off[2][2];
  unsigned a_mask[2][2];
  int b_off[2][2];
  unsigned b_ok = 0;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gi = hh * 16 + wave + 8 * j;
      const int m = m0 + gi * 8 + lrow;
      a_mask[hh][j] = 0;
      a_off[hh][j] = 0;
      if (AMODE == 0) {
        a_off[hh][j] = (int)((long long)m * a.lda + kc_of(gi));
        a_mask[hh][j] = m < a.M ? 1u : 0u;
      } else if (m < a.M) {
        const int wm = m % g.Wm;
        const int t = m / g.Wm;
        const int hm = t % g.Hm;
        const int ni = t / g.Hm;
        const int bh = g.mode == 1 ? hm * g.sh - g.ph : hm + g.ph;
        const int bw = g.mode == 1 ? wm * g.sw - g.pw : wm + g.pw;
        a_off[hh][j] = (int)((long long)ni * g.Hg * g.Wg * g.ldg + ((long long)bh * g.Wg + bw) * g.ldg + kc_of(gi));
        unsigned msk = 0;
        for (int r = 0; r < g.R; ++r)
          for (int s2 = 0; s2 < g.S; ++s2) {
            int hi, wi;
            if (conv_src(g, hm, wm, r, s2, hi, wi)) msk |= 1u << (r * g.S + s2);
          }
        a_mask[hh][j] = msk;
      }
      // B: LDS row -> output channel permutation inside a 64-column wave tile (see igemm_nt_glds_kernel)
      const int row = gi * 8 + lrow;
      const int rb = row % 64;
      const int n = n0 + (row - rb) + ((rb >> 2) & 3) * 16 + (rb >> 4) * 4 + (rb & 3);
      if (n < a.N) b_ok |= 1u << (hh * 2 + j);
      b_off[hh][j] = (int)((long long)n * a.ldb + kc_of(gi));
    }

  const int sgn = g.mode == 1 ? 1 : -1;
  const bool k_exact = (a.K % BK) == 0;
  const int nk = (a.K + BK - 1) / BK;
  // scalar state of the NEXT K-tile of A to stage (AMODE 2: tap walk; AMODE 0: k offset)
  int s_tap = 0, s_r = 0, s_s = 0, s_c0 = 0, s_ka = 0;

  auto stage_a = [&](int S, int hh) {   // one A half-tile of the K-tile described by the scalar state
    uint4* dst = smem + S * STAGE_VECS;
    if (AMODE == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int gi = hh * 16 + wave + 8 * j;
        const bool ok = a_mask[hh][j] && (k_exact || s_ka + kc_of(gi) < a.K);
        glds16_buf(rsA, ok, a_off[hh][j] + s_ka, 1, dst + gi * 64);
      }
    } else {
      const int koff = sgn * ((s_r * g.dh * g.Wg + s_s * g.dw) * g.ldg) + s_c0;
      const unsigned bit = 1u << s_tap;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int gi = hh * 16 + wave + 8 * j;
        glds16_buf(rsA, a_mask[hh][j] & bit, a_off[hh][j] + koff, 1, dst + gi * 64);
      }
    }
  };
  auto advance_a = [&]() {
    if (AMODE == 0) {
      s_ka += BK;
    } else {
      s_c0 += BK;
      if (s_c0 >= g.Cg) {
        s_c0 = 0;
        ++s_tap;
        if (++s_s == g.S) { s_s = 0; ++s_r; }
      }
    }
  };
  auto stage_b = [&](int S, int hh, int k0) {
    uint4* dst = smem + S * STAGE_VECS + BM * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gi = hh * 16 + wave + 8 * j;
      const bool ok = ((b_ok >> (hh * 2 + j)) & 1u) && (k_exact || k0 + kc_of(gi) < a.K);
      glds16_buf(rsB, ok, b_off[hh][j] + k0, 1, dst + gi * 64);
    }
  };

  const int wm_ = wave & 1, wn_ = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int xsw = (l15 >> 1) & 7;
  // fragment base offsets (vector index inside a stage): row*8 + (kc ^ xsw), kc = ks*4 + lg
  const int fa_base = (wm_ * 128 + l15) * 8 + (lg ^ xsw);
  const int fb_base = BM * 8 + (wn_ * 64 + l15) * 8 + (lg ^ xsw);

  f32x4 acc[4][8];   // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Frag fa0[4][2], fa1[4][2], fb0[2][2], fb1[2][2];   // [tile][ks]
  auto read_a = [&](const uint4* st, int mh, Frag (&f)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) f[j][ks] = *(const Frag*)&st[(ks ? (fa_base ^ 4) : fa_base) + (mh * 4 + j) * 128];
  };
  auto read_b = [&](const uint4* st, int nh, Frag (&f)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) f[i][ks] = *(const Frag*)&st[(ks ? (fb_base ^ 4) : fb_base) + (nh * 2 + i) * 128];
  };
  auto mma_quad = [&](int mh, int nh, const Frag (&fa)[4][2], const Frag (&fb)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<T>::run(acc[nh * 2 + i][mh * 4 + j], fb[i][ks], fa[j][ks]);
  };

  // ---- prologue: K-tile 0 (all four half-tiles) and the A half-tiles of K-tile 1
  stage_b(0, 0, 0);
  stage_b(0, 1, 0);
  stage_a(0, 0);
  stage_a(0, 1);
  advance_a();
  if (nk > 1) {
    stage_a(1, 0);
    stage_a(1, 1);
    advance_a();
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  // one K-tile; S = its LDS stage (compile-time in the unrolled-by-2 loop below)
  auto ktile = [&](int c, int S) {
    const uint4* st = smem + S * STAGE_VECS;
    const bool more1 = (ABL & 1) ? false : c + 1 < nk, more2 = (ABL & 1) ? false : c + 2 < nk;
    const bool rd = (ABL & 2) ? c == 0 : true;           // ablation: read the fragments of K-tile 0 only
    // ---- q0
    if (rd) {
    read_a(st, 0, fa0);
    read_b(st, 0, fb0);
    read_b(st, 1, fb1);                                  // prefetch for q1
    }
    if (more1) stage_b(S ^ 1, 0, (c + 1) * BK);          // BH0 of K-tile c+1 (overwrites BH0 of K-tile c-1)
    MR_P8_SCHED();
    mma_quad(0, 0, fa0, fb0);
    MR_P8_SCHED();
    // ---- q1
    if (rd) read_a(st, 1, fa1);                          // prefetch for q2
    if (more1) stage_b(S ^ 1, 1, (c + 1) * BK);          // BH1 of K-tile c+1
    MR_P8_SCHED();
    mma_quad(0, 1, fa0, fb1);
    MR_P8_SCHED();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of AH(c) are retired
    __builtin_amdgcn_s_barrier();
    // ---- q2
    if (rd) read_b(st, 0, fb0);                          // B0 again, for q3
    if (more2) stage_a(S, 0);                            // AH0 of K-tile c+2 (overwrites AH0 of K-tile c)
    MR_P8_SCHED();
    mma_quad(1, 1, fa1, fb1);
    MR_P8_SCHED();
    // ---- q3
    if (more2) { stage_a(S, 1); advance_a(); }           // AH1 of K-tile c+2
    MR_P8_SCHED();
    mma_quad(1, 0, fa1, fb0);
    MR_P8_SCHED();
    if (more2)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K-tile c+1 landed; AH0/AH1 of c+2 stay in flight
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  int c = 0;
  for (; c + 1 < nk; c += 2) {
    ktile(c, 0);
    ktile(c + 1, 1);
  }
  if (c < nk) ktile(c, 0);

  EpiColStats<bf16_t, 4> cst;
  bool with_stats = false;
  if constexpr (EpiHasStats<Epi>::value) with_stats = epi.stats != nullptr;
  if (with_stats) cst.init();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x4 run[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) run[i] = acc[i][j];
    if constexpr (EpiHasStats<Epi>::value)
      if (with_stats) cst.add(epi, m0 + wm_ * 128 + j * 16 + l15, n0 + wn_ * 64 + lg * 16, run);
    epi.template store_run<4>(m0 + wm_ * 128 + j * 16 + l15, n0 + wn_ * 64 + lg * 16, run);
  }
  if constexpr (EpiHasStats<Epi>::value)
    if (with_stats) cst.flush(epi, n0 + wn_ * 64 + lg * 16, l15);
}

}  // namespace mr
