// NT implicit-GEMM kernel v5 ("ping-pong"): 8 waves, v_mfma_f32_32x32x16_bf16, bf16 operands.
//
// Same operands, same LDS image ([rows][8 chunks x 16 B], physical chunk = kc ^ ((row>>1)&7)), same LDS-DMA staging and the
// same results as igemm_nt_big_kernel (igemm_core.h).  What changes (round 6):
//
//  * MFMA shape.  The accumulator block of a wave is TM x TN blocks of 32 x 32 (16 registers each); one instruction does
//    32 x 32 x 16 = twice the flops of a 16x16x32 for the same operand registers, issues at 32 cycles per SIMD and leaves 5
//    free issue slots beside it (MI355X_MICROARCH.md, "hidden per v_mfma_f32_32x32x16_bf16 gap").  A-fragment of a 32-row
//    block at k16-substep kk: lane l reads row l & 31, chunk 2 kk + (l >> 5) -- with the XOR swizzle above the four
//    16-lane service groups of the ds_read_b128 each touch 16 distinct 16-byte bank slots (rows {0-3,12-15,20-27} have
//    distinct (row & 1, (row >> 1) & 7)), so the existing image is conflict-free for this shape too.
//  * Schedule.  The eight waves are two GROUPS of four (one wave per SIMD each: waves w and w + 4 share a SIMD).  A k-tile
//    is PH sub-steps; a wave alternates a MEMORY phase (all fragment reads of a sub-step into registers; the LDS-DMA of a
//    later k-tile) and a COMPUTE phase (nothing but the MFMAs of that sub-step), a raw s_barrier after every phase, and the
//    groups run ONE PHASE APART: while group 0 computes, group 1 reads -- the matrix pipe of every SIMD always has one wave in
//    its compute phase (the v3 kernel and the phased v4 kernel run both waves of a SIMD in lock step: 43 % of their wave
//    cycles are spent parked at s_waitcnt / s_barrier with the matrix pipe 35 % busy, profiles/r02_pmc_sq_pass1_v1.txt).
//
//    slot:        0        1        2            3        4            5
//    group 0:   M(0)     C(0)     M(1)+dma(2)  C(1)     M(2)+dma(3)  C(2) ...
//    group 1:   --       M(0)     C(0)+dma(2)  M(1)     C(1)+dma(3)  M(2) ...          (PH = 1)
//
//    Two LDS stages (k-tile t lives in stage t & 1).  Stage t & 1 is last read in slot 2 PH t + 2 PH - 1 (group 1), so
//    k-tile t + 2 is staged by BOTH groups in slot 2 PH (t + 1) -- group 0 from its memory phase, group 1 between the MFMAs of
//    its compute phase -- and first read in slot 2 PH (t + 2): each wave drains its own LDS-DMA (s_waitcnt vmcnt(0)) at the end
//    of slot 2 PH (t + 2) - 1, one k-tile of MFMAs after it issued them.
//    Hazards: RAW -- vmcnt(0) of the issuing wave + the barrier closing the slot; WAR -- every wave retires its fragment reads
//    (s_waitcnt lgkmcnt(0)) before the barrier that closes its memory phase, the overwriting DMA is issued behind that barrier.
//
// Output mapping: weights are the MFMA's A operand, so lane l holds pixel (column) l & 31 and, in accumulator register
// r = 4 q + e, channel row 8 q + 4 (l >> 5) + e of the 32-channel block.  The B rows are permuted when they are staged
// (LDS row b*32 + q*8 + h*4 + e of a wave tile holds channel h*16*TN + b*16 + q*4 + e) so that each lane owns 16 TN CONSECUTIVE
// channels of its pixel: 16-byte stores, 32 TN bytes per lane and row.
#pragma once
#include "igemm_core.h"

namespace mr {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Column statistics for the 32x32 accumulator layout (same contract as EpiColStats, forward mode): a lane owns 16 TN
// consecutive channels of its TM pixel rows, the 32 lanes of a half-wave (same l >> 5) own the same channels.  Done four
// channels at a time (8 live registers beside the accumulators): sums over the lane's rows, DPP + one cross-row exchange
// over the 32 lanes, lanes 0-3 add the sums and lanes 4-7 the sums of squares.
template <typename Epi>
__device__ __forceinline__ void nt32_stats4(const Epi& epi, int n, int l31, const float (&s4)[4], const float (&q4)[4]) {
  float sv[4], qv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sv[e] = row16_sum(s4[e]);
    qv[e] = row16_sum(q4[e]);
    sv[e] += __shfl_xor(sv[e], 16);
    qv[e] += __shfl_xor(qv[e], 16);
  }
  const int c = l31 & 3;
  float v = l31 < 4 ? sv[0] : qv[0];
#pragma unroll
  for (int e = 1; e < 4; ++e) v = c == e ? (l31 < 4 ? sv[e] : qv[e]) : v;
  double* dst = epi.stats + (size_t)(blockIdx.x % epi.stats_ncopy) * 2 * epi.N;
  if (l31 < 8 && n + c < epi.N) atomicAdd(dst + (l31 < 4 ? 0 : epi.N) + n + c, (double)v);
}

// WM x WN = 8 waves; wave tile = (TM * 32) x (TN * 32); PH = sub-steps per 64-deep k-tile (1, 2 or 4).
// OPT bit 0: s_setprio 1 around the compute phase.  OPT bit 1: group 1 issues its LDS-DMA in front of its MFMAs instead of
// between them.  ABL (tools build only, wrong results): 1 = no LDS-DMA in the loop, 2 = no fragment reads in the loop.
template <int WM, int WN, int TM, int TN, int AMODE, typename Epi, int PH = 1, int OPT = 0, int ABL = 0>
__global__ __launch_bounds__(512) void igemm_nt32_kernel(NtArgs a, ConvGeom g, Epi epi) {
  typedef bf16_t T;
  typedef bf16x8 Frag;
  static_assert(WM * WN == 8, "igemm_nt32_kernel: 8 waves");
  static_assert(PH == 1 || PH == 2 || PH == 4, "igemm_nt32_kernel: 1, 2 or 4 sub-steps per k-tile");
  static_assert(AMODE == 0 || AMODE == 2, "igemm_nt32_kernel: dense or fast conv gather only");
  constexpr int VEC = 8, BK = 64, NW = 8;
  constexpr int WTM = TM * 32, WTN = TN * 32, BM = WM * WTM, BN = WN * WTN;
  constexpr int AG = BM / 8, BG = BN / 8;                      // 8-row staging groups
  constexpr int AI = (AG + NW - 1) / NW, BI = (BG + NW - 1) / NW;
  constexpr int TILE_VECS = (BM + BN) * 8;
  constexpr int KS = 4 / PH;                                   // k16 sub-steps per phase

  extern __shared__ uint4 smem_nt32[];
  uint4* smem = smem_nt32;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M - a.m_begin + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;      // XCD-aware map, see igemm_nt_big_kernel
  const int tile_m = (slot / tiles_n) * 8 + xcd, tile_n = slot % tiles_n;
  if (tile_m >= tiles_m) return;
  const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;
  const int lrow = lane >> 3, lpc = lane & 7;

  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;
  (void)a.zero;
  const rsrc_t rsA = make_rsrc(A), rsB = make_rsrc(B);

  // ---- staging descriptors (as igemm_nt_big_kernel): 32-bit element offset of (row, logical chunk) at k = 0 / tap (0,0)
  auto kc_of = [&](int gi) { return (lpc ^ (((gi * 8 + lrow) >> 1) & 7)) * VEC; };
  int a_off[AI];
  unsigned a_mask[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int gi = wave + i * NW;
    const int m = m0 + gi * 8 + lrow;
    a_mask[i] = 0;
    a_off[i] = 0;
    if (gi < AG) {
      if (AMODE == 0) {
        a_off[i] = (int)((long long)m * a.lda + kc_of(gi));
        a_mask[i] = m < a.M ? 1u : 0u;
      } else if (m < a.M) {
        const int wm = m % g.Wm;
        const int t = m / g.Wm;
        const int hm = t % g.Hm;
        const int ni = t / g.Hm;
        const int bh = g.mode == 1 ? hm * g.sh - g.ph : hm + g.ph;
        const int bw = g.mode == 1 ? wm * g.sw - g.pw : wm + g.pw;
        a_off[i] = (int)((long long)ni * g.Hg * g.Wg * g.ldg + ((long long)bh * g.Wg + bw) * g.ldg + kc_of(gi));
        unsigned msk = 0;
        for (int r = 0; r < g.R; ++r)
          for (int s2 = 0; s2 < g.S; ++s2) {
            int hi, wi;
            if (conv_src(g, hm, wm, r, s2, hi, wi)) msk |= 1u << (r * g.S + s2);
          }
        a_mask[i] = msk;
      }
    }
  }
  int b_off[BI];
  unsigned b_okmask = 0;
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int gi = wave + i * NW;
    const int row = gi * 8 + lrow;
    const int rb = row % WTN;   // LDS row -> channel permutation inside a wave tile (header comment)
    const int n = n0 + (row - rb) + ((rb >> 2) & 1) * (16 * TN) + (rb >> 5) * 16 + ((rb >> 3) & 3) * 4 + (rb & 3);
    if (gi < BG && n < a.N) b_okmask |= 1u << i;
    b_off[i] = (int)((long long)n * a.ldb + kc_of(gi));
  }

  const int sgn = g.mode == 1 ? 1 : -1;
  const bool k_exact = (a.K % BK) == 0;
  const int nk = (a.K + BK - 1) / BK;
  // scalar state of the NEXT k-tile this wave stages; s_t = its index.  Pieces of k-tiles >= nk are still ISSUED (no branch in
  // the loop) with every lane's offset out of range: the buffer resource returns zeros without touching memory.
  int s_tap = 0, s_r = 0, s_s = 0, s_c0 = 0, s_k0 = 0, s_t = 0;

  // one LDS-DMA piece (1 KiB) of that k-tile: pieces 0 .. AI-1 = A groups, AI .. AI+BI-1 = B groups
  auto stage_piece = [&](uint4* sA, int p) {
    const bool live = (ABL & 1) ? s_t < 2 : s_t < nk;
    if (p < AI) {
      const int gi = wave + p * NW;
      if (AG % NW == 0 || gi < AG) {
        if (AMODE == 0) {
          const bool ok = live & (a_mask[p] != 0) & (k_exact | (s_k0 + kc_of(gi) < a.K));
          glds16_buf(rsA, ok, a_off[p] + s_k0, 1, sA + gi * 64);
        } else {
          const int koff = sgn * ((s_r * g.dh * g.Wg + s_s * g.dw) * g.ldg) + s_c0;
          const bool ok = live & (((a_mask[p] >> (s_tap & 31)) & 1u) != 0);
          glds16_buf(rsA, ok, a_off[p] + koff, 1, sA + gi * 64);
        }
      }
    } else {
      const int i = p - AI;
      const int gi = wave + i * NW;
      if (BG % NW == 0 || gi < BG) {
        const bool ok = live & (((b_okmask >> i) & 1u) != 0) & (k_exact | (s_k0 + kc_of(gi) < a.K));
        glds16_buf(rsB, ok, b_off[i] + s_k0, 1, sA + BM * 8 + gi * 64);
      }
    }
  };
  auto stage_advance = [&]() {
    ++s_t;
    s_k0 += BK;
    if (AMODE == 2) {
      s_c0 += BK;
      if (s_c0 >= g.Cg) {
        s_c0 = 0;
        ++s_tap;
        if (++s_s == g.S) { s_s = 0; ++s_r; }
      }
    }
  };
  auto stage_all = [&](uint4* sA) {
#pragma unroll
    for (int p = 0; p < AI + BI; ++p) stage_piece(sA, p);
    stage_advance();
  };

  const int wm_ = wave % WM, wn_ = wave / WM;
  // ping-pong group: the two waves of a SIMD must be in DIFFERENT groups.  OPT bit 2 = 0: waves w and w + 4 are assumed to
  // share a SIMD (grp = wave >> 2); OPT bit 2 = 1: waves 2 i and 2 i + 1 (grp = wave & 1)
  const int grp = (OPT & 4) ? (wave & 1) : (wave >> 2);
  const int l31 = lane & 31, hl = lane >> 5;
  const int xsw = (l31 >> 1) & 7;
  // fragment base offsets (vector index inside a stage) per k16 sub-step kk: row*8 + ((2 kk + hl) ^ xsw)
  int fa_b[4], fb_b[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    fa_b[kk] = (wm_ * WTM + l31) * 8 + ((2 * kk + hl) ^ xsw);
    fb_b[kk] = BM * 8 + (wn_ * WTN + l31) * 8 + ((2 * kk + hl) ^ xsw);
  }

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  Frag fa[KS][TM], fb[KS][TN];
  auto read_frags = [&](const uint4* st, int p) {
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int kk = p * KS + q;
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[q][i] = *(const Frag*)&st[fb_b[kk] + i * 256];
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[q][j] = *(const Frag*)&st[fa_b[kk] + j * 256];
    }
  };
  // the MFMAs of one phase.  DMA: the LDS-DMA pieces of this wave's next k-tile go to stage `sd`, spread between the MFMAs
  // (piece p behind MFMA number ceil((p + 1) NM / (NP + 1))) or, OPT bit 1, in front of them.
  auto compute = [&](auto dma_tag, uint4* sd) {
    constexpr bool DMA = decltype(dma_tag)::value;
    constexpr int NP = AI + BI;
    constexpr int NM = KS * TN * TM;
    if (DMA && (OPT & 2)) {
#pragma unroll
      for (int p = 0; p < NP; ++p) stage_piece(sd, p);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (OPT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < KS; ++q)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[q][i], fa[q][j], acc[i][j], 0, 0, 0);
          if (DMA && !(OPT & 2)) {
            const int im = (q * TN + i) * TM + j + 1;
#pragma unroll
            for (int p = 0; p < NP; ++p)
              if (im == ((p + 1) * NM + NP) / (NP + 1)) {
                __builtin_amdgcn_sched_barrier(0);
                stage_piece(sd, p);
                __builtin_amdgcn_sched_barrier(0);
              }
          }
        }
    if (OPT & 1) __builtin_amdgcn_s_setprio(0);
    if (DMA) stage_advance();
  };
  typedef std::true_type Yes;
  typedef std::false_type No;

  uint4* const st0 = smem;
  uint4* const st1 = smem + TILE_VECS;

  // ---- prologue: k-tile 0 by everybody; k-tile 1 by group 1 here (its first slot is idle), by group 0 in its phase M(0, 0)
  stage_all(st0);
  if (grp == 1) stage_all(st1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  if (grp == 0) {
    // group 0: M(t, p) in slot 2 (PH t + p), C(t, p) in the next; stages k-tile t + 1 (other stage) from M(t, 0) and drains it
    // behind C(t, PH - 1)
    auto ktile = [&](int t, uint4* cur, uint4* oth) {
#pragma unroll
      for (int p = 0; p < PH; ++p) {
        if (!(ABL & 2) || t == 0) read_frags(cur, p);
        if (p == 0) stage_all(oth);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        compute(No{}, cur);
        __builtin_amdgcn_sched_barrier(0);
        if (p == PH - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      ktile(t, st0, st1);
      ktile(t + 1, st1, st0);
    }
    if (t < nk) ktile(t, st0, st1);
  } else {
    // group 1: one slot later; stages k-tile t + 2 (this stage) between the MFMAs of C(t, PH - 1), drains it behind M(t + 1, PH - 1)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    auto ktile = [&](int t, uint4* cur, uint4* oth) {
      (void)oth;
#pragma unroll
      for (int p = 0; p < PH; ++p) {
        if (!(ABL & 2) || t == 0) read_frags(cur, p);
        if (p == PH - 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (p == PH - 1) compute(Yes{}, cur);
        else compute(No{}, cur);
        __builtin_amdgcn_sched_barrier(0);
        // (no barrier behind the very last phase: it pairs with group 0's missing first one)
        if (p < PH - 1 || t + 1 < nk) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      ktile(t, st0, st1);
      ktile(t + 1, st1, st0);
    }
    if (t < nk) ktile(t, st0, st1);
  }

  // ---- epilogue: lane (hl, l31) owns channels n0 + wn_*WTN + hl*16*TN .. +16*TN-1 of pixel rows m0 + wm_*WTM + j*32 + l31
  constexpr int NC = 16 * TN;
  const int nb = n0 + wn_ * WTN + hl * NC;
  if constexpr (EpiHasStats<Epi>::value) {
    if (epi.stats != nullptr) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + b * 16 + q * 4;
          float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm_ * WTM + j * 32 + l31;
            if (m < epi.M) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float t = acc[b][j][4 * q + e];
                if (epi.bias && n + e < epi.N) t += epi.bias[n + e];
                t = to_f32(from_f32<bf16_t>(t));   // the value the store writes
                s4[e] += t;
                q4[e] += t * t;
              }
            }
          }
          nt32_stats4(epi, n, l31, s4, q4);
        }
    }
  }
  if ((ABL & 4) && epi.M >= 0) {   // timing-only: no epilogue stores (one lane keeps the accumulators alive)
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[b][j][e];
    if (t == 123.456f) epi.C[0] = from_f32<bf16_t>(t);
    return;
  }
  // Output through LDS (OPT bit 3 clear; whole 16-byte channel vectors, no residual addend): the tile is written to LDS as
  // bf16 [BM][BN] (16-byte chunk c of row r at chunk c ^ (r & (CH-1)): conflict-free both ways), then stored row-major, 64
  // lanes = 1 KiB = whole 128-byte lines per store instruction.  The direct stores below put 64 16-byte pieces of 32 different
  // rows into one instruction: measured 15-16 us for the 33.5 MB of a 65536 x 256 output against 7.2 us for a plain fill
  // (tools/probe_nt_fixed_cost.py, profiles/r06_nt_fixed_cost.txt).
  constexpr int CH = BN / 8;
  static_assert((CH & (CH - 1)) == 0, "output tile width: a power of two of 16-byte chunks");
  const bool via_lds = !(OPT & 8) && epi.vec_ok && (epi.ldc % 8) == 0 && (epi.N % 8) == 0 && epi.addend == nullptr;
  if (via_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pieces of k-tiles >= nk (zeros) have landed too
    __syncthreads();                                    // every wave is done with the stage buffers
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int r = wm_ * WTM + j * 32 + l31;
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int n = nb + b * 16 + qq * 8;
          bf16_t o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = acc[b][j][qq * 8 + e];
            if (epi.bias && n + e < epi.N) t += epi.bias[n + e];
            if (epi.relu) t = fmaxf(t, 0.f);
            o[e] = from_f32<bf16_t>(t);
          }
          const int c = (wn_ * WTN + hl * NC) / 8 + b * 2 + qq;
          smem[r * CH + (c ^ (r & (CH - 1)))] = *(const uint4*)o;
        }
    }
    __syncthreads();
    bf16_t* __restrict__ Cp = epi.C;
    // every workgroup starts at another row of its tile: the 256 workgroups finish their k-loops together, and tiles that all
    // start at row 0 put their first lines 128 KiB apart -- on the same few memory channels at every instant
    const int rot = (OPT & 16) ? 0 : (int)((blockIdx.x * 8u) % (unsigned)BM);
#pragma unroll 4
    for (int idx = tid; idx < BM * CH; idx += 512) {
      int r = idx / CH + rot;
      if (r >= BM) r -= BM;
      const int c = idx % CH;
      const int m = m0 + r, n = n0 + c * 8;
      if (m < epi.M && n < epi.N) *(uint4*)(Cp + (long long)m * epi.ldc + n) = smem[r * CH + (c ^ (r & (CH - 1)))];
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    f32x4 run[4 * TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        run[b * 4 + q] = (f32x4){acc[b][j][4 * q], acc[b][j][4 * q + 1], acc[b][j][4 * q + 2], acc[b][j][4 * q + 3]};
    epi.template store_run<4 * TN>(m0 + wm_ * WTM + j * 32 + l31, nb, run);
  }
}

}  // namespace mr
