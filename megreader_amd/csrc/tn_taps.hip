// All-taps weight-gradient kernel (bf16) for 3x3, stride 1, padding == dilation convolutions.
//
// Replaces (on MI355X) cuDNN's wgrad behind nn.Conv2d.backward for the 3x3 layers of the CRNN backbone and the ResNet
// bottlenecks (reference: backbones/crnn.py:44-55, backbones/resnet.py:39-56,110-140).
//
// Why: the 128x128 TN kernel (igemm_tn_glds_kernel) treats wgrad as a plain GEMM  dw[Cout, 9*Cin] = dy^T * im2col(x),
// so every tile stages the x rows once PER TAP -- 8 LDS-DMA pieces per wave per 32 MFMAs, and the DMA issue cost is
// what bounds it (DESIGN.md "What bounds the igemm kernels").  Here one workgroup owns a (64 Cout) x (64 Cin) block for
// ALL nine taps: the x rows are staged once into an LDS ring and each tap reads them at a shifted row, so a wave
// issues 4 pieces per 72 MFMAs.
//
// The pixel stream.  The reduction runs over a 1-D stream of positions q = n*IP + y*Wp + x with Wp = W + d (d zero
// columns after every row, d = dilation = padding) and IP = H*Wp rounded up to 8.  dys[q] / xs[q] are dy / x at real
// pixels and 0 at pad positions.  Then, for tap (r, s):   dw[:, r, s, :] = sum_q  m_r(q) * dys[q] (x) xs[q + shift],
// shift = (r-1)*d*Wp + (s-1)*d.  Horizontal out-of-image taps land on the zero pad columns; vertical ones would land
// in the neighbouring image, so dys is masked per 8 positions for r = 0 (rows y < d) and r = 2 (rows y >= H-d):
// m_r is periodic in q with period IP and is kept as a small LDS table of AND masks.
// A caller-owned table (one int per stream position: pixel index or -1) turns the stream into source offsets.
//
// LDS: x ring  = 2^RL chunks of [64 positions][64 channels] (128-byte rows; 32-byte pieces XOR-swizzled by
//                h(row) = bit1(row) | bit3(row) << 1, which makes every ds_read_b64_tr_b16 -- at ANY row shift --
//                hit 8 distinct 32-byte slots of the 256-byte bank row per 32-lane group),
//      dy ring = 4 chunks of the same shape, mask table = IP/8 entries of 32 bytes.
// Step c (one 64-position chunk): barrier; issue the LDS-DMA of chunk c+HALO+1; 2 x (4 dy fragments, 9 x fragments,
// 36 MFMAs) per wave.  Wave w owns Cin columns 16w..16w+15 of the block for all 4 Cout row tiles and all 9 taps
// (36 accumulator tiles = 144 registers).
#include "tn_taps.h"
#include "tuning.h"
#include "igemm_core.h"

namespace mr {

struct TapArgs {
  const void* A;   // dy
  const void* B;   // x
  float* C;
  const int* tab;
  float* colsum;
  int NA, Cg, lda, ldg, ldc;
  int Wp, IP8;          // row pitch of the stream, image pitch / 8
  int top_lo, bot_hi;   // r = 0 valid  <=>  (q mod IP) >= top_lo ;   r = 2 valid  <=>  (q mod IP) < bot_hi
  int dil;
  int nchunks;          // table length / 64
  int cps;              // chunks per split
  // Group reduction of the split partials (grp > 1): the workgroups of `grp` consecutive splits of a tile write their
  // accumulators to slabs in `ws`, take a ticket, and the last arriver sums the group's slabs into its registers and is
  // the only one that issues the f32 atomics.  ws = [4096 int tickets (zero between launches)][gridDim slabs x 147456 B].
  int grp;
  void* ws;
  // fin != 0 (needs grp > 1): the group leaders do NOT add into dw; they leave the group's sum in the slab of the group's
  // first split and taps_finalize_kernel (a second, tiny launch over ALL the CUs) adds the groups' slabs into dw with plain
  // read-modify-writes.  Why: the leaders' f32 atomics are issued per lane (~1 per clock per CU) and cost ~20 us of tail
  // during which the other CUs idle.
  int fin;
};

constexpr int TAPS_TICKETS = 4096;
constexpr long long TAPS_SLAB_BYTES = 36ll * 256 * 16;

// stream table: tab[q] = pixel index (n*H + y)*W + x, or -1 at pad positions / beyond the batch
static __global__ void taps_table_kernel(int N, int H, int W, int Wp, int IP, int len, int* __restrict__ tab) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= len) return;
  const int n = q / IP, rem = q - n * IP;
  const int y = rem / Wp, x = rem - y * Wp;
  tab[q] = (n < N && y < H && x < W) ? (n * H + y) * W + x : -1;
}

__device__ __forceinline__ int taps_hash(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

// LDS addresses are plain byte offsets: the kernel has no static LDS, so its dynamic allocation starts at 0.  (Going
// through the `extern __shared__` symbol makes hipcc add a link-time "+0" to every computed address.)
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
__device__ __forceinline__ bf16x8 tr_read2(int p0, int p1) {
  const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds_byte_t*)(uintptr_t)(unsigned)p0);
  const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds_byte_t*)(uintptr_t)(unsigned)p1);
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ void taps_glds16(rsrc_t r, bool ok, int elem_off, int lds_wave_base) {
  const unsigned voff = ok ? ((unsigned)elem_off << 1) : 0xFFFFFFFFu;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(lds_byte_t*)(uintptr_t)(unsigned)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 and_mask(bf16x8 v, u32x4 m) {
  return __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, v) & m);
}

// RL: log2 of the x ring length in chunks (2 -> 32 KB, 3 -> 64 KB).  HALO: chunks of x needed either side of the
// current one (|shift| <= 64*HALO).  COLSUM: the tile_b == 0 workgroups also produce the bias gradient.
// ABL (timing only, wrong results): bit 0 = no LDS-DMA in the loop, bit 1 = x fragments read once per chunk half and
// re-used for every tap, bit 2 = no edge masks, bit 3 = no atomic epilogue, bit 4 = no barrier in the loop.
// W8: 8-wave workgroup, one per CU: waves 0-3 and 4-7 are two independent copies of the 4-wave kernel (own LDS rings,
// consecutive split ranges) that share the barriers; at the end waves 4-7 hand their accumulators to waves 0-3 through
// LDS, so a CU produces ONE partial tile instead of two (the partials -- 147 KB per workgroup -- are what the epilogue
// costs: their volume is (#workgroups) x (tile size), whatever the reduction scheme).
template <int RL, int HALO, int ABL = 0, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256, W8 ? 1 : 2) void igemm_tn_taps_kernel(TapArgs a) {
  constexpr int ROWB = 128, CHB = 64 * ROWB;                 // bytes per row / per 64-row chunk
  constexpr int XRING = (1 << RL) * CHB, XMASK = XRING - 1;  // x ring bytes
  constexpr int ARING = 4 * CHB;
  constexpr int NT = W8 ? 512 : 256;
  constexpr int sM = (W8 ? 2 : 1) * (XRING + ARING);  // mask table: IP8 entries x {top[4], bottom[4]} dwords

  const int tid8 = threadIdx.x;
  const int khalf = W8 ? __builtin_amdgcn_readfirstlane(tid8 >> 8) : 0;   // which K-half of the workgroup
  const int tid = tid8 & 255;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int sX = khalf * (XRING + ARING), sA = sX + XRING;   // byte offsets of this half's rings
  const int tiles_b = a.Cg >> 6, tiles_a = (a.NA + 63) >> 6;
  // XCD-aware work map, as igemm_tn_glds_kernel: split-major virtual order, every XCD takes a contiguous eighth, so
  // the workgroups that share an L2 stream through the same rows of dy / x.
  const int total = gridDim.x;
  const int xq = total >> 3, xr = total & 7, xcd = blockIdx.x & 7;
  const int vb = xcd * xq + (xcd < xr ? xcd : xr) + (blockIdx.x >> 3);
  const int ntiles = tiles_a * tiles_b;
  const int split = vb / ntiles, tile = vb - split * ntiles;   // split = the WORKGROUP's index along the reduction
  const int tile_b = tile % tiles_b, tile_a = tile / tiles_b;
  const int na0 = tile_a * 64, cb0 = tile_b * 64;
  // every (half-)workgroup runs exactly cps steps; chunks beyond the table stage zeros (uniform barrier counts)
  if (split * (W8 ? 2 : 1) * a.cps >= a.nchunks) return;
  const int c_begin = (split * (W8 ? 2 : 1) + khalf) * a.cps;
  const int c_end = c_begin + a.cps;

  // ---- staging: this lane moves, per chunk, rows piece*8 + srow of pieces wave*2 + {0, 1}, physical 16-byte chunk
  // lane & 7 of the 128-byte row.  The swizzle is applied to the SOURCE column (LDS-DMA writes lane-linearly).
  const int srow = lane >> 3, pslot = (lane & 7) >> 1, half = lane & 1;
  int scol[2], srowc[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    srowc[jj] = (wave * 2 + jj) * 8 + srow;
    scol[jj] = (((pslot ^ taps_hash(srowc[jj])) << 1) + half) * 8;
  }
  const bool okA0 = na0 + scol[0] < a.NA, okA1 = na0 + scol[1] < a.NA;
  const rsrc_t rsA = make_rsrc(a.A), rsB = make_rsrc(a.B);
  int ent[2];
  auto fetch_entries = [&](int cc) {
    const bool in = cc >= 0 && cc < a.nchunks;  // uniform
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) ent[jj] = in ? a.tab[cc * 64 + srowc[jj]] : -1;
  };
  auto stage = [&](int cc, bool with_a) {
    const int dX = sX + ((cc & ((1 << RL) - 1)) * CHB) + wave * 2048;
    const int dA = sA + ((cc & 3) * CHB) + wave * 2048;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const bool ok = ent[jj] >= 0;
      taps_glds16(rsB, ok, ent[jj] * a.ldg + cb0 + scol[jj], dX + jj * 1024);
      if (with_a) taps_glds16(rsA, ok && (jj ? okA1 : okA0), ent[jj] * a.lda + na0 + scol[jj], dA + jj * 1024);
    }
  };

  // ---- fragment read addresses
  const int l15 = lane & 15, lg = lane >> 4;
  const int frow = lg * 8 + (l15 >> 2);  // row this lane supplies (hh = 0), before kk / chunk / tap shift
  int offA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) offA[i] = frow * ROWB + ((i ^ taps_hash(frow)) << 5) + (l15 & 3) * 8;
  // (frow + 4) has the same bit 1 and bit 3 as frow when bit 2 of frow is clear; frow = lg*8 + (0..3): always clear.
  int o0[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int shift = ((t / 3) - 1) * a.dil * a.Wp + ((t % 3) - 1) * a.dil;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int r0 = (frow + hh * 4 + shift) & ((64 << RL) - 1);
      o0[t][hh] = r0 * ROWB + ((wave ^ taps_hash(r0)) << 5) + (l15 & 3) * 8;
    }
  }
  // mask-table entry of this lane's 8 positions (kk = 0 of the first chunk); advances by 4 entries per 32 positions
  int ment = (int)(((long long)c_begin * 8 + lg) % a.IP8);
  const int ment_step = 4 % a.IP8;

  const bool do_colsum = a.colsum != nullptr && tile_b == 0;
  f32x4 acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 accs[4];   // column sums of dy through the MFMA pipe (B = ones): only wave 0 of the tile_b == 0 workgroups
#pragma unroll
  for (int i = 0; i < 4; ++i) accs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;
  const bool cs_wave = do_colsum && wave == 0;   // uniform

  // ---- prologue: mask table, first HALO+... chunks
  for (int e = tid8; e < a.IP8; e += NT) {
    u32x4 mt, mb;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int q0 = e * 8 + 2 * d, q1 = q0 + 1;
      mt[d] = (q0 >= a.top_lo ? 0xFFFFu : 0u) | (q1 >= a.top_lo ? 0xFFFF0000u : 0u);
      mb[d] = (q0 < a.bot_hi ? 0xFFFFu : 0u) | (q1 < a.bot_hi ? 0xFFFF0000u : 0u);
    }
    *(__attribute__((address_space(3))) u32x4*)(lds_byte_t*)(uintptr_t)(unsigned)(sM + e * 32) = mt;
    *(__attribute__((address_space(3))) u32x4*)(lds_byte_t*)(uintptr_t)(unsigned)(sM + e * 32 + 16) = mb;
  }
  {   // all the prologue's table entries first (one L2 round trip), then its LDS-DMAs
    int pe[2 * HALO + 1][2];
#pragma unroll
    for (int cc = -HALO; cc <= HALO; ++cc) {
      fetch_entries(c_begin + cc);
      pe[cc + HALO][0] = ent[0];
      pe[cc + HALO][1] = ent[1];
    }
    fetch_entries(c_begin + HALO + 1);
    const int e0 = ent[0], e1 = ent[1];
#pragma unroll
    for (int cc = -HALO; cc <= HALO; ++cc) {
      ent[0] = pe[cc + HALO][0];
      ent[1] = pe[cc + HALO][1];
      stage(c_begin + cc, cc >= 0 && c_begin + cc < c_end);
    }
    ent[0] = e0;
    ent[1] = e1;
  }

  for (int c = c_begin; c < c_end; ++c) {
    if (!(ABL & 16)) __syncthreads();
    if (!(ABL & 1)) {
      stage(c + HALO + 1, c + HALO + 1 < c_end);
      fetch_entries((ABL & 2048) ? c_begin : c + HALO + 2);   // ABL 2048, timing only: always the same (L2-hot) rows
    }
    const int bA = sA + (c & 3) * CHB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kofs = ((c * 2 + kk) & ((2 << RL) - 1)) << 12;  // 32 rows = 4096 bytes; uniform
      bf16x8 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa[i] = tr_read2(bA + offA[i] + kk * 32 * ROWB, bA + offA[i] + kk * 32 * ROWB + 4 * ROWB);
      const u32x4 mt = *(const __attribute__((address_space(3))) u32x4*)(lds_byte_t*)(uintptr_t)(unsigned)(sM + ment * 32);
      const u32x4 mb = *(const __attribute__((address_space(3))) u32x4*)(lds_byte_t*)(uintptr_t)(unsigned)(sM + ment * 32 + 16);
      ment += ment_step;
      if (ment >= a.IP8) ment -= a.IP8;
      if (cs_wave) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accs[i], 0, 0, 0);
      }
      // x fragments one tap ahead of the MFMAs that consume them; the vertical-edge masks are applied to the x
      // fragment (same reduction index as dy's: 4 registers per tap instead of 16 for the dy row tiles)
      bf16x8 fb_next = tr_read2(sX + ((o0[0][0] + kofs) & XMASK), sX + ((o0[0][1] + kofs) & XMASK));
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        bf16x8 fb = fb_next;
        if (t + 1 < 9 && !(ABL & 2))
          fb_next = tr_read2(sX + ((o0[t + 1][0] + kofs) & XMASK), sX + ((o0[t + 1][1] + kofs) & XMASK));
        if (t < 3 && !(ABL & 4)) fb = and_mask(fb, mt);
        if (t >= 6 && !(ABL & 4)) fb = and_mask(fb, mb);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][t], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: atomic accumulation into dw (f32)
  if (ABL & 8) {
    float keep = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) keep += acc[i][t][0] + acc[i][t][1] + acc[i][t][2] + acc[i][t][3];
    if (keep == 123.456f) a.C[0] = keep;
    return;
  }
  if (W8) {   // waves 4-7 -> LDS -> waves 0-3 (register order, 16 bytes per lane and tile; the rings are dead)
    __syncthreads();
    if (khalf == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          *(__attribute__((address_space(3))) f32x4*)(lds_byte_t*)(uintptr_t)(unsigned)(((i * 9 + t) * 256 + tid) * 16) = acc[i][t];
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[i][t] += *(const __attribute__((address_space(3))) f32x4*)(lds_byte_t*)(uintptr_t)(unsigned)(((i * 9 + t) * 256 + tid) * 16);
    }
  }
  const bool worker = khalf == 0;   // the threads that own the workgroup's partial tile from here on
  if (cs_wave && l15 == 0) {   // every column of (dy^T * ones) holds the same sums
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = na0 + i * 16 + lg * 4 + q;
        if (row < a.NA) atomicAdd(a.colsum + row, accs[i][q]);
      }
  }
  if (a.grp > 1 || a.fin == 2) {
    // In-launch reduction over a group of splits (no spinning: correct for any residency / dispatch order).  Slabs are in
    // "register order" -- element (k, tid) of a slab is acc tile k of thread tid -- so writer and reader use the same
    // coalesced 16-byte accesses.  Publish = sc1 (write-through) slab stores, wait, barrier, then a relaxed agent-scope
    // ticket; the last arriver reads the slabs with sc1 loads (cdna guide, split-K reducer recipe, sc1 form: no
    // release / acquire fences -- a release would write back the whole XCD L2, i.e. everybody's slabs, once per
    // workgroup: measured no faster than the atomics it replaced).
    const int nsplits = total / ntiles;
    const int g0 = (split / a.grp) * a.grp;
    const int gsize = a.fin == 2 ? 2 : min(a.grp, nsplits - g0);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char*)a.ws + TAPS_TICKETS * 4, (short)0, 0x7fffffff,
                                                       0x00020000);
    if (gsize > 1) {
      int* tickets = (int*)a.ws;
      const int so = vb * (int)TAPS_SLAB_BYTES;
      if (worker) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 9; ++t)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][t]), rs, tid * 16,
                                                   so + (i * 9 + t) * 4096, 16);
      }
      // fin == 2: no in-launch reduction at all -- the partial tile stays in this workgroup's slab (no ticket, no wait) and
      // taps_finalize_kernel, the next launch, sums the splits of a tile into dw
      if (a.fin == 2) return;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      volatile __attribute__((address_space(3))) int* bc = (volatile __attribute__((address_space(3))) int*)(lds_byte_t*)(uintptr_t)0u;
      int* tk = tickets + (tile * ((nsplits + a.grp - 1) / a.grp) + split / a.grp) % TAPS_TICKETS;
      if (tid8 == 0) *bc = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int ticket = *bc;
      if (ticket != gsize - 1) return;
      if (tid8 == 0) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      for (int j = 0; worker && j < gsize && !(ABL & 512); ++j) {   // ABL 512, timing only: no slab reads
        const int vbj = (g0 + j) * ntiles + tile;
        if (vbj == vb) continue;
        const int sj = vbj * (int)TAPS_SLAB_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 9; ++t)
            acc[i][t] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, sj + (i * 9 + t) * 4096, 16));
      }
    }
    if (a.fin) {   // leave the group's sum in the slab of its first split; taps_finalize_kernel adds it into dw
      if (worker) {
        const int sg = (g0 * ntiles + tile) * (int)TAPS_SLAB_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 9; ++t)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][t]), rs, tid * 16,
                                                   sg + (i * 9 + t) * 4096, 16);
      }
      return;
    }
  }
  if (!worker) return;
  if (ABL & 1024) return;   // timing only: no final atomics (group phase only)
  if ((ABL & 128) && wave >= 2) return;              // timing only: half the waves issue atomics
  if ((ABL & 256) && (blockIdx.x >> 3) & 1) return;   // timing only: half the workgroups issue atomics
#pragma unroll
  for (int t = 0; t < ((ABL & 64) ? 3 : 9); ++t) {   // ABL 64, timing only: a third of the atomics
    const int col = t * a.Cg + cb0 + wave * 16 + l15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = na0 + i * 16 + lg * 4 + q;
        if (row < a.NA) {
          if (ABL & 32) a.C[(long long)row * a.ldc + col] = acc[i][t][q];   // timing only: plain stores
          else atomicAdd(a.C + (long long)row * a.ldc + col, acc[i][t][q]);
        }
      }
  }
}

// dw += sum over the groups of their slab (see TapArgs.fin).  One workgroup per (tile, accumulator tile k = i*9 + t);
// thread tid of the slab order = thread tid of the producing workgroup, so its f32x4 is rows lg*4 .. +3 of column
// (wave*16 + l15) of the 16 x 64 block (i, t).
static __global__ __launch_bounds__(256) void taps_finalize_kernel(const f32x4* __restrict__ slabs, float* __restrict__ C,
                                                                  int ntiles, int tiles_b, int nsplits, int grp, int NA,
                                                                  int Cg, int ldc) {
  const int tile = blockIdx.x / 36, k = blockIdx.x - tile * 36;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int g0 = 0; g0 < nsplits; g0 += grp)
    sum += slabs[((long long)g0 * ntiles + tile) * (36 * 256) + k * 256 + tid];
  const int tile_b = tile % tiles_b, tile_a = tile / tiles_b;
  const int i = k / 9, t = k - i * 9;
  const int col = t * Cg + tile_b * 64 + wave * 16 + l15;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = tile_a * 64 + i * 16 + lg * 4 + q;
    if (row < NA) C[(long long)row * ldc + col] += sum[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static int taps_num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
      cus = p.multiProcessorCount;
    else
      cus = 256;
  }
  return cus;
}

struct TapsLayout {
  int Wp, IP, len;   // len = table length (multiple of 64)
};
static TapsLayout taps_layout(int N, int H, int W, int dil) {
  TapsLayout l;
  l.Wp = W + dil;
  l.IP = (H * l.Wp + 7) / 8 * 8;
  l.len = (int)(((long long)N * l.IP + 63) / 64 * 64);
  return l;
}

int taps_eligible(int N, int H, int W, int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph,
                  int pw, int dh, int dw, int Ho, int Wo, long long tab_bytes) {
  if (R != 3 || S != 3 || sh != 1 || sw != 1 || dh != dw || ph != dh || pw != dw || Ho != H || Wo != W) return 0;
  if (dh < 1 || dh >= H || dh >= W) return 0;
  if ((long long)N * H * W < MR_TUNE(tn_taps_min_p)) return 0;   // few output pixels: the plain TN GEMM kernel wins (round 4 A/B)
  if (Cin % 64 != 0 || ldx % 8 != 0 || lddy % 8 != 0 || Cout % 8 != 0) return 0;
  const long long stream = (long long)N * ((H * (W + dh) + 7) / 8 * 8);
  if (stream + 64 >= (1ll << 31) / 4) return 0;
  const TapsLayout l = taps_layout(N, H, W, dh);
  if (dh * l.Wp + dh > 64) return 0;                 // |tap shift| <= one chunk (HALO = 1)
  if (l.IP < 32 || l.IP / 8 * 32 > 4096) return 0;   // mask table: 4 KB of LDS
  if ((long long)l.len * 4 > tab_bytes) return 0;
  if ((long long)N * H * W * ldx * 2 >= (1ll << 31) || (long long)N * H * W * lddy * 2 >= (1ll << 31)) return 0;
  return 1;
}

// one workspace per device (one process may drive several GPUs): indexed by the current device at set / launch time
static void* g_taps_ws_dev[64] = {nullptr};
static long long g_taps_ws_bytes_dev[64] = {0};
#define g_taps_grp MR_TUNE(tn_taps_group)   // 0 = automatic, 1 = atomics only, > 1 = forced group size
static int taps_cur_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  return dev;
}
// set while a launch may run concurrently with other TN launches on another stream (mr_conv2d_wgrad_tab flags bit 1): the shared
// split-reduction workspace (tickets + slabs) must then not be used -- its launches have to be stream-ordered with each other
static thread_local bool g_tn_concurrent = false;
void taps_set_concurrent(bool on) { g_tn_concurrent = on; }
void taps_set_workspace(void* p, long long bytes) {
  const int dev = taps_cur_dev();
  if (dev < 0) return;
  g_taps_ws_dev[dev] = p;
  g_taps_ws_bytes_dev[dev] = p ? bytes : 0;
}
void taps_get_workspace(void** p, long long* bytes) {
  const int dev = taps_cur_dev();
  *p = (dev >= 0 && !g_tn_concurrent) ? g_taps_ws_dev[dev] : nullptr;
  *bytes = dev >= 0 ? g_taps_ws_bytes_dev[dev] : 0;
}

// 1: 8-wave workgroups (two reduction halves share one partial tile), one per CU.  Opt-in: it halves the partial-tile
// traffic (HBM-side bytes per launch 209 -> 144 MB) and ties in the microbenchmark, but inside the training step it is
// ~10 % slower (rocprofv3 average 122.6 vs 111.2 us per launch: one barrier-coupled 8-wave workgroup per CU)
#define g_taps_w8 MR_TUNE(tn_taps_w8)

// How the split partials of a tile reach dw (mr_tuning.tn_taps_fin):
//   0 (default): groups of splits reduce in the launch (slabs + tickets), the group leaders add into dw with f32 atomics;
//   1: same groups, but the leaders leave the group sums in their slabs and a finalize launch adds them into dw -- measured
//      equal to 0 (conv2..5 wgrad 445 vs 445 us per step at group 4: the second launch + 19 MB of slab reads cost what the
//      leaders' atomics cost);
//   2: no in-launch reduction at all -- every workgroup stores its partial tile to its own slab (no tickets, no waits) and the
//      finalize launch sums ALL the splits of a tile.  The kernel itself gets 5 % shorter (109 -> 104 us per launch on the CRNN
//      layers = 0.34 of peak), the step does not follow: CRNN 2.835 -> 2.81 ms, FPN-attention 10.12 -> 10.00 ms, but Res50-PPM
//      12.90 -> 13.03 ms (16 more launches per step reading 4-8 split slabs each).  Opt-in.
//   (A first version of mode 2 with its own copy of the slab stores pushed this 254-VGPR kernel into scratch -- 180 bytes per
//   lane -- and slowed modes 0 / 1 from 109 to 155 us per launch; the modes share one store sequence now, resource usage is
//   what it was.)
#define g_taps_fin MR_TUNE(tn_taps_fin)

#ifdef MR_ABLATION
static int g_taps_abl = 0;
int taps_set_abl(int mask) { const int old = g_taps_abl; g_taps_abl = mask; return old; }
#endif

int launch_tn_taps(const TapsProblem& p, int splits_override, hipStream_t stream) {
  const TapsLayout l = taps_layout(p.N, p.H, p.W, p.dil);
  if (p.build) {
    hipLaunchKernelGGL(taps_table_kernel, dim3(cdiv(l.len, 256)), dim3(256), 0, stream, p.N, p.H, p.W, l.Wp, l.IP,
                       l.len, p.tab);
    MR_CHECK_LAUNCH();
  }
  TapArgs a;
  a.A = p.dy; a.B = p.x; a.C = p.dw; a.tab = p.tab; a.colsum = p.dbias;
  a.NA = p.Cout; a.Cg = p.Cin; a.lda = p.lddy; a.ldg = p.ldx; a.ldc = 9 * p.Cin;
  a.Wp = l.Wp; a.IP8 = l.IP / 8;
  a.top_lo = p.dil * l.Wp;
  a.bot_hi = (p.H - p.dil) * l.Wp;
  a.dil = p.dil;
  a.nchunks = l.len / 64;
  const int tiles = cdiv(p.Cout, 64) * (p.Cin / 64);
  const int cus = taps_num_cus();
  const int w8 = g_taps_w8;
  const int per_cu = w8 ? 1 : 2;   // resident workgroups per CU
  const int halves = w8 ? 2 : 1;   // independent reduction ranges per workgroup
  // split count (workgroups along the reduction): a workgroup pays ~3 chunks of prologue and a 147 KB epilogue
  int splits = 1;
  double best = 1e300;
  for (int s = 1; s * halves <= a.nchunks; ++s) {
    const long long blocks = (long long)tiles * s;
    const long long rounds = (blocks + per_cu * cus - 1) / (per_cu * cus);
    const double cost = (double)rounds * (cdiv(a.nchunks, s * halves) + 12.0);
    if (cost < best) { best = cost; splits = s; }
    if (blocks > 8ll * cus) break;
  }
  if (splits_override > 0) splits = splits_override;
  if (splits * halves > a.nchunks) splits = a.nchunks / halves > 0 ? a.nchunks / halves : 1;
  a.cps = cdiv(a.nchunks, splits * halves);
  splits = cdiv(a.nchunks, a.cps * halves);
  // group reduction: needs the registered workspace (tickets + one slab per workgroup) and unique tickets
  a.grp = 1;
  const int dev = taps_cur_dev();
  void* const g_taps_ws = (dev >= 0 && !g_tn_concurrent) ? g_taps_ws_dev[dev] : nullptr;
  const long long g_taps_ws_bytes = dev >= 0 ? g_taps_ws_bytes_dev[dev] : 0;
  a.ws = g_taps_ws;
  {
    int want = g_taps_grp ? g_taps_grp : (splits >= 4 ? 4 : (splits >= 2 ? 2 : 1));
    if (want > splits) want = splits;
    const long long need = TAPS_TICKETS * 4ll + (long long)tiles * splits * TAPS_SLAB_BYTES;
    if (want > 1 && g_taps_ws && need <= g_taps_ws_bytes && need < (1ll << 31) &&
        (long long)tiles * cdiv(splits, want) <= TAPS_TICKETS)
      a.grp = want;
  }
  a.fin = (a.grp > 1 && g_taps_fin == 1) ? 1 : 0;
  if (g_taps_fin == 2 && splits > 1 && !w8) {   // direct slabs + finalize launch (see the kernel's fin == 2 branch)
    const long long need = TAPS_TICKETS * 4ll + (long long)tiles * splits * TAPS_SLAB_BYTES;
    if (g_taps_ws && need <= g_taps_ws_bytes && need < (1ll << 31)) {
      a.fin = 2;
      a.grp = 1;
    }
  }
  const int lds = w8 ? 147456 : (4 + 4) * 64 * 128 + 4096;
  const int threads = w8 ? 512 : 256;
  static bool attr_set[2] = {false, false};
  const void* kp = w8 ? (const void*)igemm_tn_taps_kernel<2, 1, 0, true> : (const void*)igemm_tn_taps_kernel<2, 1, 0, false>;
  if (!attr_set[w8]) {
    if (hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", lds);
      return MR_ERR_LAUNCH;
    }
    attr_set[w8] = true;
  }
#ifdef MR_ABLATION
  if (g_taps_abl && !w8) {
    a.fin = 0;
#define MR_TAPS_ABL(V_) case V_: { auto k2 = igemm_tn_taps_kernel<2, 1, V_>; \
      (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
      hipLaunchKernelGGL(k2, dim3(tiles * splits), dim3(256), lds, stream, a); } break;
    switch (g_taps_abl) { MR_TAPS_ABL(1) MR_TAPS_ABL(2) MR_TAPS_ABL(4) MR_TAPS_ABL(8) MR_TAPS_ABL(16) MR_TAPS_ABL(6)
      MR_TAPS_ABL(7) MR_TAPS_ABL(15) MR_TAPS_ABL(31) MR_TAPS_ABL(32) MR_TAPS_ABL(64) MR_TAPS_ABL(128) MR_TAPS_ABL(256) MR_TAPS_ABL(512) MR_TAPS_ABL(1024) MR_TAPS_ABL(1536) MR_TAPS_ABL(2048) MR_TAPS_ABL(2056) MR_TAPS_ABL(9)
      default: break; }
#undef MR_TAPS_ABL
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
#endif
  if (w8)
    hipLaunchKernelGGL((igemm_tn_taps_kernel<2, 1, 0, true>), dim3(tiles * splits), dim3(threads), lds, stream, a);
  else
    hipLaunchKernelGGL((igemm_tn_taps_kernel<2, 1, 0, false>), dim3(tiles * splits), dim3(threads), lds, stream, a);
  MR_CHECK_LAUNCH();
  if (a.fin) {
    hipLaunchKernelGGL(taps_finalize_kernel, dim3(tiles * 36), dim3(256), 0, stream,
                       (const f32x4*)((const char*)a.ws + TAPS_TICKETS * 4), a.C, tiles, p.Cin / 64, splits,
                       a.fin == 2 ? 1 : a.grp, a.NA, a.Cg, a.ldc);
    MR_CHECK_LAUNCH();
  }
  return MR_OK;
}

}  // namespace mr
