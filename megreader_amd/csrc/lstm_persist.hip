// Persistent bidirectional-LSTM recurrence for gfx950: ONE launch per layer and pass instead of one dependent
// launch per time step (reference: cuDNN RNN behind nn.LSTM at decoders/crnn.py:13,21,91-93).
//
// Why: at CRNN shapes (T=33, N=256, H=256) a step is 134 MFLOP per direction -- microseconds of MFMA work -- and the
// 132 dependent step launches of one training step cost 5.7-6.6 us each (launch boundary + refetch of W_hh from L2).
//
// Decomposition (bf16, H = 256).  Samples are independent through the recurrence, hidden units are not:
//   * the batch is cut into groups of 16 rows (one MFMA M-tile): groups never talk to each other;
//   * inside a group, the recurrent matrix is cut into G = 4 slices, one workgroup (4 waves) per slice.  A slice is
//     128 KB of bf16 = 32 MFMA fragments per lane: it is loaded ONCE and stays in the wave's VGPRs for all T steps
//     (W_hh never touches LDS, L2 or HBM again);
//   * both directions run concurrently (independent chains): grid = 2 x ceil(N/16) x 4 workgroups (128 at N=256).
//   forward : slice g owns the gate columns of hidden units [64g, 64g+64).  h_t[16, 256] is ALL-GATHERED between the
//             4 slices: every lane publishes its h values as 8-byte {bf16 pair, tag} granules with write-through
//             (sc1) stores; consumers sweep the granules with sc1 loads until every tag equals the step number -- the
//             data is its own flag (no fence, no separate flag round trip).  12 KB in per workgroup per step.
//   backward: dh_{t-1} = dgates_t . W_hh has its REDUCTION dimension (the 1024 gate columns) distributed, so slice g
//             multiplies its own 256 gate columns (which it just computed, no gather) by W_hh[own cols, all units] and
//             the f32 partial sums are REDUCE-SCATTERED: wave w's output tile belongs to slice w; {f32, tag} granules,
//             24 KB in per workgroup per step (an all-gather of dgates would be 48 KB).
//   Cell state c_t (forward) and the carried dc (backward) live in registers; cbuf / gates / out are written as the
//   step kernels write them, so the weight-gradient GEMMs that follow are unchanged.
//
// Inter-workgroup protocol (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility",
// cdna_hip_programming.md Guideline 16 form R2): granule = one naturally aligned 8-byte {value, tag} written by ONE
// relaxed agent-scope atomic store (global_store_dwordx2 sc1) and read by relaxed agent-scope atomic loads
// (global_load_dwordx2 sc1): placement-independent, no dependence on dispatch order beyond co-residency of the 4
// members of a group.  tag = step + 1 (never 0); the exchange buffer is zeroed by a memset node ahead of every launch;
// two slots alternate (a producer can run at most one step ahead of its slowest consumer).  Every spin is bounded:
// on timeout the workgroup records a code in the status word, stops waiting (no hang) and POISONS its outputs with NaN, so
// the failure reaches the loss / the gradients instead of silently corrupting a training run.
#include "common.h"
#include "igemm_core.h"
#include "../../include/megreader_hip.h"

namespace mr {

constexpr int PH = 256;        // hidden size the persistent kernels are built for
constexpr int PG = 4;          // slices (workgroups) per batch group
constexpr int PR = 16;         // batch rows per group (one MFMA M tile)
constexpr int PHS = PH / PG;   // hidden units per slice (64)
constexpr int PLD = PH + 8;    // LDS row stride (elements): 16 rows x 16-byte fragment reads without bank conflicts
constexpr unsigned SPIN_LIMIT = 1u << 21;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int FWD_GRAN = PR * PH / 2;                 // granules per slot, forward  (bf16 pair per granule): 2048
constexpr int BWD_GRAN_SLAB = PR * PHS;               // granules per (dest, src) slab, backward (one f32 each): 1024
constexpr int BWD_GRAN = PG * (PG - 1) * BWD_GRAN_SLAB;  // per slot: 12288

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int AUX_SC1 = 16;   // cache-policy bit of the raw buffer builtins: sc1 = agent scope (write-through / L1 bypass)

// Two adjacent granules travel as ONE 16-byte sc1 access: {value0, tag, value1, tag}.  Each 8-byte half is a granule on
// its own (carries its tag), so the pair needs no atomicity beyond the naturally aligned 8 bytes.
__device__ __forceinline__ void gran2_store(rsrc_t r, unsigned byte_off, unsigned v0, unsigned v1, unsigned tag) {
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, tag, v1, tag}, r, (int)byte_off, 0, AUX_SC1);
}
__device__ __forceinline__ u32x4 gran2_load(rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX_SC1);
}
// `local` (uniform): all four slices of the batch group were found on ONE XCD (xcd_colocated below).  A plain store
// then KEEPS the line in that XCD's L2 and the siblings' sc1 (L1-bypassing) polls hit it there; an sc1 store drops
// the line from L2 and every poll pays the fabric round trip (MI355X_MICROARCH.md, "stores of each flavour").
__device__ __forceinline__ void gran2_publish(rsrc_t r, unsigned byte_off, unsigned v0, unsigned v1, unsigned tag,
                                              bool local) {
  if (local)
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, tag, v1, tag}, r, (int)byte_off, 0, 0);
  else
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, tag, v1, tag}, r, (int)byte_off, 0, AUX_SC1);
}

constexpr unsigned HELLO_TAG = 0x48454C4Fu;
constexpr unsigned HELLO_SPINS = 1u << 16;

// Workgroup -> (slice g, batch group bg, direction).  xcd_map: hardware workgroup b is dispatched to XCD b % 8
// (round-robin), so the four slices of a group take block indices that are congruent mod 8.
__device__ __forceinline__ void persist_roles(int xcd_map, int nbg, int& g, int& bg, int& dir) {
  const int b = blockIdx.x;
  if (xcd_map) {
    const int r = b >> 3, dg = (r >> 2) * 8 + (b & 7);
    g = r & 3;
    dir = dg / nbg;
    bg = dg - dir * nbg;
  } else {
    g = b % PG;
    bg = (b / PG) % nbg;
    dir = b / (PG * nbg);
  }
}

// Are the four slices of this batch group on one XCD?  The placement above is a dispatch-order ASSUMPTION, so it is
// verified: every workgroup publishes its HW_REG_XCC_ID (sc1 store: visible anywhere) and reads its three siblings'.
// Only if all four agree does this workgroup publish with plain stores.  A sibling that does not answer within the
// spin bound counts as "elsewhere" (sc1 stores are always correct).  `sh` is one LDS word.
__device__ __forceinline__ bool xcd_colocated(rsrc_t rx, unsigned hello_base, int g, int xcd_map, int* sh,
                                              unsigned* status) {
  if (!xcd_map) return false;
  unsigned me;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(me));
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) gran2_store(rx, hello_base + (unsigned)(g * 16), me, me, HELLO_TAG);
  if (tid < 64) {
    bool same = true;
    if (lane < PG - 1) {
      const int gf = lane + (lane >= g);
      same = false;
      for (unsigned spins = 0; spins < HELLO_SPINS; ++spins) {
        const u32x4 v = gran2_load(rx, hello_base + (unsigned)(gf * 16));
        if (v[1] == HELLO_TAG) { same = v[0] == me; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    const bool all_same = __all(same);
    if (lane == 0) {
      *sh = all_same ? 1 : 0;
      if (all_same) atomicAdd(status + 1, 1u);
    }
  }
  __syncthreads();
  return *sh != 0;
}

// Gate non-linearities on the dependency chain of the recurrence: v_exp_f32 + v_rcp_f32 forms (a few ulp; this path
// only exists in bf16 compute mode, where h is rounded to 8 mantissa bits right after).  Saturate correctly:
// exp -> inf gives rcp -> 0.
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16_t x = (bf16_t)a, y = (bf16_t)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

struct LstmPFwd {
  const bf16_t* xproj;  // [T*N, 8H]  (dir-major, gate-interleaved)
  const bf16_t* whh;    // [2][4H][H] gate-interleaved rows
  bf16_t* out;          // [T, N, 2H]
  float* cbuf;          // [T, N, 2H]
  bf16_t* gates;        // [T, N, 8H]
  u64* xch;             // [2 dirs][nbg][2 slots][FWD_GRAN]
  unsigned* status;
  int T, N, nbg;
  int xcd_map;          // block -> role map that puts a group's slices on one XCD (see persist_roles)
  unsigned hello_off;   // byte offset (from xch) of the XCC-id exchange: [2*nbg][PG] granule pairs
};

// Ownership inside a slice (64 hidden units, 256 gate columns, 4 waves x 4 MFMA tiles): tile i of wave w takes the
// gate columns {4 * (16w + 4*(t/4) + i) + t%4 : t = 0..15} as its 16 rows, so that the MFMA output lane (lg, l15)
// holds, over i = 0..3, all four gates of the four CONSECUTIVE units 16w + 4lg + i of batch row l15: the lane's
// x-projection is 32 contiguous bytes, its gates 32, its cell states 16, its h values 8 -- and its two granules one
// 16-byte store.
__global__ __launch_bounds__(256, 1) void lstm_fwd_persist_kernel(LstmPFwd a) {
  typedef Mma<bf16_t>::Frag Frag;
  __shared__ __attribute__((aligned(16))) bf16_t hbuf[2][PR][PLD];
  __shared__ int local_sh;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  int g, bg, dir;
  persist_roles(a.xcd_map, a.nbg, g, bg, dir);
  const int row = bg * PR + l15;          // batch row of this lane's accumulator column
  const bool row_ok = row < a.N;
  constexpr int H = PH;

  // ---- the slice of W_hh this wave multiplies, as MFMA fragments held in registers for the whole sequence
  Frag wf[4][8];
  {
    const bf16_t* wbase = a.whh + (long long)dir * 4 * H * H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 4 * (g * PHS + wave * 16 + (l15 >> 2) * 4 + i) + (l15 & 3);   // gate column (row of whh)
#pragma unroll
      for (int c = 0; c < 8; ++c) wf[i][c] = *(const Frag*)(wbase + (long long)n * H + c * 32 + lg * 8);
    }
  }
  const rsrc_t rx = make_rsrc(a.xch);
  const unsigned xbase = (unsigned)(((long long)dir * a.nbg + bg) * 2 * FWD_GRAN * 8);   // bytes
  const bool local = xcd_colocated(rx, a.hello_off + (unsigned)((dir * a.nbg + bg) * PG * 16), g, a.xcd_map, &local_sh,
                                   a.status);
  const int u0 = wave * 16 + lg * 4;          // first of this lane's 4 units inside the slice
  const int j0 = g * PHS + u0;                // ... as a hidden-unit index
  float cst[4] = {0.f, 0.f, 0.f, 0.f};
  bool dead = false;
  // The step's global stores (out / cbuf / gates, 56 bytes a lane) are off the chain, but a vector-memory wait counts loads and
  // stores together: issued right behind the publish they made the NEXT step's granule sweep wait for their write
  // acknowledgements.  They are held in registers and issued when that sweep has completed, in the shadow of the MFMAs.
  uint2 pend_h = make_uint2(0, 0);
  f32x4 pend_c = {0.f, 0.f, 0.f, 0.f};
  uint4 pend_g0 = make_uint4(0, 0, 0, 0), pend_g1 = pend_g0;
  long long pend_r = -1;
  // phase clock (tools/microbench_lstm.py --phases): thread 0 of workgroup 0 adds the 100 MHz wall clock spent in each phase of a
  // step into status words [8 + phase]; only when the caller set status word 2 (the product never does)
  const bool timing = tid == 0 && blockIdx.x == 0 && a.status[2] == 0x54494D45u;
  unsigned long long tprev = timing ? wall_clock64() : 0ull;
  unsigned tacc[4] = {0, 0, 0, 0};
#define LSTM_TICK(i)                                  \
  if (timing) {                                       \
    const unsigned long long tn_ = wall_clock64();    \
    tacc[i] += (unsigned)(tn_ - tprev);               \
    tprev = tn_;                                      \
  }

  // The x-projection (32 bytes a lane out of a 34 MB buffer: HBM latency) is fetched ONE STEP AHEAD, right behind the granule sweep,
  // and the barrier behind it orders LDS only: issued in front of the sweep the loads made the sweep wait for them (loads return in
  // order), behind it a __syncthreads() drained them.  116 -> 101 us per layer forward at N = 256 together with the held-back stores.
  uint4 nx0, nx1;       // (rows beyond the batch read row N - 1: their results are never stored)
  {
    const bf16_t* xp = a.xproj + ((long long)(dir == 0 ? 0 : a.T - 1) * a.N + min(row, a.N - 1)) * 8 * H + dir * 4 * H + 4 * j0;
    nx0 = *(const uint4*)xp;
    nx1 = *(const uint4*)(xp + 8);
  }
  for (int s = 0; s < a.T; ++s) {
    const int t = dir == 0 ? s : a.T - 1 - s;
    const long long r = (long long)t * a.N + row;
    // operands of the gate math that do not depend on the recurrent term: in flight during the exchange.  (Fetching them a step
    // ahead -- behind the sweep, behind the MFMAs or at the end of the step -- measured 2-5 us slower per layer: the loads
    // then sit in front of the fragment reads or the gate math instead of under the hand-off wait.)
    const uint4 x0 = nx0, x1 = nx1;
    // UNCONDITIONAL loads (row and step clamped into the buffer): a load under a divergent or uniform branch is waited for at the
    // join -- s_waitcnt vmcnt right behind it, the whole latency on the chain (seen in the ISA of the backward kernel)
    const int sn = min(s + 1, a.T - 1);
    const bf16_t* xpn = a.xproj + ((long long)(dir == 0 ? sn : a.T - 1 - sn) * a.N + min(row, a.N - 1)) * 8 * H + dir * 4 * H + 4 * j0;
    if (s == 0) { nx0 = *(const uint4*)xpn; nx1 = *(const uint4*)(xpn + 8); }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      // ---- all-gather h_{prev}: this thread fetches, from each of the 3 foreign slices, the granule pair of the
      // thread with its own (wave, lane) -- tag s, slot (s-1)&1.  The own slice went through LDS (below).
      bf16_t* hb = &hbuf[s & 1][0][0];
      u32x4 v[3];
      const unsigned src = xbase + (unsigned)(((s - 1) & 1) * FWD_GRAN * 8) + (unsigned)(tid * 16);
      if (!dead) {
        for (unsigned spins = 0;; ++spins) {
          bool ok = true;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int gf = k + (k >= g);
            v[k] = gran2_load(rx, src + (unsigned)(gf * 256 * 16));
            ok &= v[k][1] == (unsigned)s && v[k][3] == (unsigned)s;
          }
          if (__all(ok)) break;
          if (spins > SPIN_LIMIT) {
            if (lane == 0) atomicMax(a.status, 1u);
            dead = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      LSTM_TICK(0)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int gf = k + (k >= g);
        *(uint2*)(hb + l15 * PLD + gf * PHS + u0) = make_uint2(v[k][0], v[k][2]);
      }
      if (pend_r >= 0) {       // the previous step's outputs
        *(uint2*)(a.out + pend_r * 2 * H + dir * H + j0) = pend_h;
        *(f32x4*)(a.cbuf + pend_r * 2 * H + dir * H + j0) = pend_c;
        bf16_t* gpp = a.gates + pend_r * 8 * H + dir * 4 * H + 4 * j0;
        *(uint4*)gpp = pend_g0;
        *(uint4*)(gpp + 8) = pend_g1;
      }
      nx0 = *(const uint4*)xpn;
      nx1 = *(const uint4*)(xpn + 8);
      lds_barrier();                           // the loads just issued stay in flight
      Frag hf[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) hf[c] = *(const Frag*)(hb + l15 * PLD + c * 32 + lg * 8);
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<bf16_t>::run(acc[i], wf[i][c], hf[c]);
    }
    LSTM_TICK(1)
    // operands of the gate math that do not depend on the recurrent term
    f32x4 xg[4];
    {
      const bf16_t* p0 = (const bf16_t*)&x0;
      const bf16_t* p1 = (const bf16_t*)&x1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xg[0][q] = (float)p0[q]; xg[1][q] = (float)p0[4 + q];
        xg[2][q] = (float)p1[q]; xg[3][q] = (float)p1[4 + q];
      }
    }
    // ---- gate math (lane owns the 4 gates of (row, unit) for 4 consecutive units), state in registers
    float hv[4], gt[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ig = fsig(acc[i][0] + xg[i][0]);
      const float fg = fsig(acc[i][1] + xg[i][1]);
      const float gg = ftanh(acc[i][2] + xg[i][2]);
      const float og = fsig(acc[i][3] + xg[i][3]);
      const float c = fg * cst[i] + ig * gg;
      cst[i] = c;
      hv[i] = og * ftanh(c);
      gt[i][0] = ig; gt[i][1] = fg; gt[i][2] = gg; gt[i][3] = og;
    }
    // a timed-out hand-off must never pass as a result: poison this workgroup's hidden state with NaN from here on, so the
    // layer output, the loss and every gradient of the step are NaN and any training loop sees it (the status
    // word alone is only read by tests)
    if (dead) {
#pragma unroll
      for (int i = 0; i < 4; ++i) hv[i] = __builtin_nanf("");
    }
    const unsigned h01 = pack_bf16(hv[0], hv[1]), h23 = pack_bf16(hv[2], hv[3]);
    if (s + 1 < a.T) {
      // publish FIRST (the hand-off is the critical path), then the own slice into the next step's LDS tile
      gran2_publish(rx, xbase + (unsigned)((s & 1) * FWD_GRAN * 8) + (unsigned)((g * 256 + tid) * 16), h01, h23,
                    (unsigned)(s + 1), local);
      *(uint2*)(&hbuf[(s + 1) & 1][l15][j0]) = make_uint2(h01, h23);
    }
    LSTM_TICK(2)
    {
      pend_h = make_uint2(h01, h23);
      pend_c = f32x4{cst[0], cst[1], cst[2], cst[3]};
      pend_g0.x = pack_bf16(gt[0][0], gt[0][1]); pend_g0.y = pack_bf16(gt[0][2], gt[0][3]);
      pend_g0.z = pack_bf16(gt[1][0], gt[1][1]); pend_g0.w = pack_bf16(gt[1][2], gt[1][3]);
      pend_g1.x = pack_bf16(gt[2][0], gt[2][1]); pend_g1.y = pack_bf16(gt[2][2], gt[2][3]);
      pend_g1.z = pack_bf16(gt[3][0], gt[3][1]); pend_g1.w = pack_bf16(gt[3][2], gt[3][3]);
      pend_r = row_ok ? r : -1;
    }
    LSTM_TICK(3)
  }
  if (pend_r >= 0) {           // the last step's outputs
    *(uint2*)(a.out + pend_r * 2 * H + dir * H + j0) = pend_h;
    *(f32x4*)(a.cbuf + pend_r * 2 * H + dir * H + j0) = pend_c;
    bf16_t* gpp = a.gates + pend_r * 8 * H + dir * 4 * H + 4 * j0;
    *(uint4*)gpp = pend_g0;
    *(uint4*)(gpp + 8) = pend_g1;
  }
  if (timing) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a.status[8 + i] = tacc[i];
  }
#undef LSTM_TICK
}

struct LstmPBwd {
  const bf16_t* dout;   // [T, N, 2H]
  const bf16_t* whhT;   // [2][H][4H]  rows = hidden unit, K = gate-interleaved column
  const float* cbuf;    // [T, N, 2H]
  bf16_t* gates;        // [T, N, 8H]  post-activation gates in, pre-activation gradients out (in place)
  u64* xch;             // [2 dirs][nbg][2 slots][BWD_GRAN]
  unsigned* status;
  int T, N, nbg;
  int xcd_map;
  unsigned hello_off;
};

__global__ __launch_bounds__(256, 1) void lstm_bwd_persist_kernel(LstmPBwd a) {
  typedef Mma<bf16_t>::Frag Frag;
  __shared__ __attribute__((aligned(16))) bf16_t abuf[PR][PLD];   // own dgates [16 rows][256 own gate columns]
  __shared__ __attribute__((aligned(16))) float obuf[4][64][4];   // own partial dh: [tile][lane][e]
  __shared__ int local_sh;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  int g, bg, dir;
  persist_roles(a.xcd_map, a.nbg, g, bg, dir);
  const int row = bg * PR + l15;
  const bool row_ok = row < a.N;
  constexpr int H = PH;

  // ---- W_hh[own 256 gate columns, all 256 units]: wave w holds the rows (units) [64w, 64w+64) of whhT
  Frag wf[4][8];
  {
    const bf16_t* wbase = a.whhT + (long long)dir * 4 * H * H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = wave * 64 + i * 16 + l15;   // hidden unit (row of whhT)
#pragma unroll
      for (int c = 0; c < 8; ++c)
        wf[i][c] = *(const Frag*)(wbase + (long long)n * 4 * H + g * 4 * PHS + c * 32 + lg * 8);
    }
  }
  const rsrc_t rx = make_rsrc(a.xch);
  const unsigned xbase = (unsigned)(((long long)dir * a.nbg + bg) * 2 * BWD_GRAN * 8);   // bytes
  const bool local = xcd_colocated(rx, a.hello_off + (unsigned)((dir * a.nbg + bg) * PG * 16), g, a.xcd_map, &local_sh,
                                   a.status);
  // epilogue ownership: thread (wave i', lane) <-> row l15, local units u0..u0+3 of this slice
  const int u0 = wave * 16 + lg * 4;
  const int j0 = g * PHS + u0;               // global hidden unit of e = 0
  float dcar[4] = {0.f, 0.f, 0.f, 0.f};
  bool dead = false;
  // the step's global store (the gate gradients, 32 bytes a lane) is held back until the NEXT step's granule sweep has completed:
  // behind the publish it made that sweep wait for its write acknowledgement (see the forward kernel)
  uint4 pend_d0 = make_uint4(0, 0, 0, 0), pend_d1 = pend_d0;
  bf16_t* pend_gp = nullptr;

  // The step's operands (dout, c_t, c_{t-1}, the saved gates: 72 bytes a lane out of 70 MB of buffers: HBM latency) are fetched
  // ONE STEP AHEAD, behind the granule sweep: issued in front of it, the sweep's wait included their latency.
  uint2 n_upr = make_uint2(0, 0);
  f32x4 n_ct = {0.f, 0.f, 0.f, 0.f}, n_cp = n_ct;
  uint4 n_x0 = make_uint4(0, 0, 0, 0), n_x1 = n_x0;
  // UNCONDITIONAL loads (row / step clamped into the buffers; c_{t-1} of the first step is masked where it is used): under a branch
  // the compiler waits for a load at the join -- `s_waitcnt vmcnt(0)` right behind the c_{t-1} load, the whole latency on the chain
  const int rowc = min(row, a.N - 1);
  auto fetch = [&](int s1) {
    const int t1 = dir == 0 ? a.T - 1 - s1 : s1;
    const int tp1 = min(max(dir == 0 ? t1 - 1 : t1 + 1, 0), a.T - 1);
    const long long r1 = (long long)t1 * a.N + rowc;
    n_upr = *(const uint2*)(a.dout + r1 * 2 * H + dir * H + j0);       // raw bf16 x 4: converted where it is used
    n_ct = *(const f32x4*)(a.cbuf + r1 * 2 * H + dir * H + j0);
    n_cp = *(const f32x4*)(a.cbuf + ((long long)tp1 * a.N + rowc) * 2 * H + dir * H + j0);
    const bf16_t* gp1 = a.gates + r1 * 8 * H + dir * 4 * H + 4 * j0;
    n_x0 = *(const uint4*)gp1;
    n_x1 = *(const uint4*)(gp1 + 8);
  };
  if (a.T > 0) fetch(0);
  // phase clock, as in the forward kernel (status words [12 + phase])
  const bool timing = tid == 0 && blockIdx.x == 0 && a.status[2] == 0x54494D45u;
  unsigned long long tprev = timing ? wall_clock64() : 0ull;
  unsigned tacc[4] = {0, 0, 0, 0};
#define LSTM_TICK(i)                                  \
  if (timing) {                                       \
    const unsigned long long tn_ = wall_clock64();    \
    tacc[i] += (unsigned)(tn_ - tprev);               \
    tprev = tn_;                                      \
  }
  for (int s = 0; s < a.T; ++s) {
    // backward visits the steps in the reverse of the forward order of that direction
    const int t = dir == 0 ? a.T - 1 - s : s;
    const long long r = (long long)t * a.N + row;
    const bool has_prev = dir == 0 ? t > 0 : t < a.T - 1;
    const f32x4 up = {__builtin_bit_cast(float, n_upr.x << 16), __builtin_bit_cast(float, n_upr.x & 0xffff0000u),
                      __builtin_bit_cast(float, n_upr.y << 16), __builtin_bit_cast(float, n_upr.y & 0xffff0000u)};
    const f32x4 ct = n_ct;
    const f32x4 cp = has_prev ? n_cp : f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4 x0 = n_x0, x1 = n_x1;
    bf16_t* gp = a.gates + r * 8 * H + dir * 4 * H + 4 * j0;
    f32x4 dh = up;
    if (s > 0) {
      // ---- reduce-scatter: 3 foreign partial sums (tag s, slot (s-1)&1) + the own one from LDS.
      // slab (dest, src index k) = 1024 granules laid out [tile 4][e-pair 2][lane 64][2]
      u32x4 v[6];
      const unsigned src = xbase + (unsigned)(((s - 1) & 1) * BWD_GRAN * 8) +
                           (unsigned)(g * (PG - 1) * BWD_GRAN_SLAB * 8) + (unsigned)((wave * 128 + lane) * 16);
      if (!dead) {
        for (unsigned spins = 0;; ++spins) {
          bool ok = true;
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int eh = 0; eh < 2; ++eh) {
              v[k * 2 + eh] = gran2_load(rx, src + (unsigned)(k * BWD_GRAN_SLAB * 8) + (unsigned)(eh * 64 * 16));
              ok &= v[k * 2 + eh][1] == (unsigned)s && v[k * 2 + eh][3] == (unsigned)s;
            }
          if (__all(ok)) break;
          if (spins > SPIN_LIMIT) {
            if (lane == 0) atomicMax(a.status, 2u);
            dead = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (pend_gp) {
        *(uint4*)pend_gp = pend_d0;
        *(uint4*)(pend_gp + 8) = pend_d1;
        pend_gp = nullptr;
      }
      const f32x4 own = *(const f32x4*)&obuf[wave][lane][0];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sum = own[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) sum += __uint_as_float(v[k * 2 + (e >> 1)][(e & 1) * 2]);
        dh[e] += sum;
      }
    }
    LSTM_TICK(0)
    // a timed-out hand-off poisons the gradients of this workgroup's rows with NaN (see the forward kernel)
    if (dead) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dh[e] = __builtin_nanf("");
    }
    f32x4 gq[4];
    {
      const bf16_t* p0 = (const bf16_t*)&x0;
      const bf16_t* p1 = (const bf16_t*)&x1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gq[0][q] = (float)p0[q]; gq[1][q] = (float)p0[4 + q];
        gq[2][q] = (float)p1[q]; gq[3][q] = (float)p1[4 + q];
      }
    }
    // ---- gate algebra (EpiLstmBwd of lstm.hip with dc carried in registers)
    float dg[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ig = gq[e][0], fg = gq[e][1], gg = gq[e][2], og = gq[e][3];
      const float tc = ftanh(ct[e]);
      const float dcv = dh[e] * og * (1.f - tc * tc) + dcar[e];
      dg[e][0] = dcv * gg * ig * (1.f - ig);
      dg[e][1] = dcv * cp[e] * fg * (1.f - fg);
      dg[e][2] = dcv * ig * (1.f - gg * gg);
      dg[e][3] = dh[e] * tc * og * (1.f - og);
      dcar[e] = dcv * fg;
    }
    uint4 d0, d1;
    d0.x = pack_bf16(dg[0][0], dg[0][1]); d0.y = pack_bf16(dg[0][2], dg[0][3]);
    d0.z = pack_bf16(dg[1][0], dg[1][1]); d0.w = pack_bf16(dg[1][2], dg[1][3]);
    d1.x = pack_bf16(dg[2][0], dg[2][1]); d1.y = pack_bf16(dg[2][2], dg[2][3]);
    d1.z = pack_bf16(dg[3][0], dg[3][1]); d1.w = pack_bf16(dg[3][2], dg[3][3]);
    if (!row_ok) d0 = d1 = make_uint4(0, 0, 0, 0);
    *(uint4*)(&abuf[l15][4 * u0]) = d0;
    *(uint4*)(&abuf[l15][4 * u0 + 8]) = d1;
    if (row_ok) {
      if (s + 1 == a.T) {
        *(uint4*)gp = d0;
        *(uint4*)(gp + 8) = d1;
      } else {
        pend_d0 = d0; pend_d1 = d1; pend_gp = gp;
      }
    }
    if (s + 1 == a.T) break;
    lds_barrier();   // abuf complete; every thread has consumed obuf of the previous step
    LSTM_TICK(1)
    // the coming step's operands: behind the gate algebra, a whole step to arrive (right behind the sweep the compiler waited for
    // them where the algebra first reads THIS step's operands: the sweep phase 1.0 -> 0.5 us, -6 us per layer with this placement;
    // refilling the operand registers in place instead of copying them at the top of the step measured slower again)
    if (s + 1 < a.T) fetch(s + 1);

    Frag af[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) af[c] = *(const Frag*)(&abuf[l15][c * 32 + lg * 8]);
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) Mma<bf16_t>::run(acc[i], wf[i][c], af[c]);
    // wave w's tile = partial dh of the units [64w, 64w+64) = slice w's units: lane holds (row l15, units 16i+4lg+e)
    if (wave == g) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(f32x4*)&obuf[i][lane][0] = acc[i];
    } else {
      // (__builtin_bit_cast on an ext-vector ELEMENT reads element 0 for every index with this compiler: the values
      // are converted through float rvalues)
      const int srcidx = g < wave ? g : g - 1;
      const unsigned dst = xbase + (unsigned)((s & 1) * BWD_GRAN * 8) +
                           (unsigned)((wave * (PG - 1) + srcidx) * BWD_GRAN_SLAB * 8) + (unsigned)(lane * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a0 = acc[i][0], a1 = acc[i][1], a2 = acc[i][2], a3 = acc[i][3];
        gran2_publish(rx, dst + (unsigned)(i * 128 * 16), __float_as_uint(a0), __float_as_uint(a1), (unsigned)(s + 1),
                      local);
        gran2_publish(rx, dst + (unsigned)((i * 128 + 64) * 16), __float_as_uint(a2), __float_as_uint(a3),
                      (unsigned)(s + 1), local);
      }
    }
    LSTM_TICK(2)
    lds_barrier();   // obuf visible; all fragment reads of abuf done before the next step overwrites it
    LSTM_TICK(3)
  }
  if (timing) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a.status[12 + i] = tacc[i];
  }
#undef LSTM_TICK
}

// layout: [granule slots][XCC-id exchange: 2*nbg groups x PG granule pairs of 16 B][status: 256 B, word 0 = timeout
// code, word 1 = number of workgroups that found their group on one XCD]
static long long persist_hello_bytes(int N) { return 2ll * cdiv(N, PR) * PG * 16; }
static long long persist_ws_bytes(int N) {
  const long long nbg = cdiv(N, PR);
  const long long fwd = 2 * nbg * 2 * FWD_GRAN * 8, bwd = 2 * nbg * 2 * (long long)BWD_GRAN * 8;
  return (fwd > bwd ? fwd : bwd) + 256 + persist_hello_bytes(N);
}

#define g_lstm_xcd (MR_TUNE(lstm_persist) != 2)   // lstm_persist = 2: never use the XCD-colocating block map (A/B knob)
// the map needs whole octets of groups: 2*nbg groups, 4 slices each, 8 XCDs
static int persist_xcd_map(int nbg) { return (g_lstm_xcd && nbg % 4 == 0) ? 1 : 0; }

// The persistent kernels need all workgroups of a batch group co-resident: keep the grid within the chip.
bool lstm_persist_ok(int dtype, int T, int N, int H) {
  return dtype == MR_BF16 && H == PH && T >= 1 && T < (1 << 30) && 2 * cdiv(N, PR) * PG <= 512;
}

int lstm_fwd_persist(const void* xproj, const void* whh, void* out, float* cbuf, void* gates, int T, int N,
                     void* ws, long long ws_bytes, hipStream_t stream) {
  const bool prezeroed = ws_bytes < 0;   // the caller zeroed it (see mr_lstm_fwd)
  if (prezeroed) ws_bytes = -ws_bytes;
  MR_CHECK_ARG(ws_bytes >= persist_ws_bytes(N), "mr_lstm_fwd: workspace too small (%lld < %lld)", ws_bytes,
               persist_ws_bytes(N));
  const int nbg = cdiv(N, PR);
  const long long xbytes = persist_ws_bytes(N) - 256 - persist_hello_bytes(N), hbytes = persist_hello_bytes(N);
  if (!prezeroed && hipMemsetAsync(ws, 0, (size_t)persist_ws_bytes(N), stream) != hipSuccess) {
    set_error("mr_lstm_fwd: memset of the exchange buffer failed");
    return MR_ERR_LAUNCH;
  }
  LstmPFwd a{(const bf16_t*)xproj, (const bf16_t*)whh, (bf16_t*)out, cbuf, (bf16_t*)gates, (u64*)ws,
             (unsigned*)((char*)ws + xbytes + hbytes), T, N, nbg, persist_xcd_map(nbg), (unsigned)xbytes};
  hipLaunchKernelGGL(lstm_fwd_persist_kernel, dim3(2 * nbg * PG), dim3(256), 0, stream, a);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

void lstm_set_bwd_debug(float*) {}   // debug hook retired with the fix of the backward publish (kept for the ABI)

int lstm_bwd_persist(const void* dout, const void* whhT, const float* cbuf, void* gates, int T, int N, void* ws,
                     long long ws_bytes, hipStream_t stream) {
  const bool prezeroed = ws_bytes < 0;   // the caller zeroed it (see mr_lstm_fwd)
  if (prezeroed) ws_bytes = -ws_bytes;
  MR_CHECK_ARG(ws_bytes >= persist_ws_bytes(N), "mr_lstm_bwd: workspace too small (%lld < %lld)", ws_bytes,
               persist_ws_bytes(N));
  const int nbg = cdiv(N, PR);
  const long long xbytes = persist_ws_bytes(N) - 256 - persist_hello_bytes(N), hbytes = persist_hello_bytes(N);
  if (!prezeroed && hipMemsetAsync(ws, 0, (size_t)persist_ws_bytes(N), stream) != hipSuccess) {
    set_error("mr_lstm_bwd: memset of the exchange buffer failed");
    return MR_ERR_LAUNCH;
  }
  LstmPBwd a{(const bf16_t*)dout, (const bf16_t*)whhT, cbuf, (bf16_t*)gates, (u64*)ws,
             (unsigned*)((char*)ws + xbytes + hbytes), T, N, nbg, persist_xcd_map(nbg), (unsigned)xbytes};
  hipLaunchKernelGGL(lstm_bwd_persist_kernel, dim3(2 * nbg * PG), dim3(256), 0, stream, a);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

long long lstm_persist_ws(int dtype, int T, int N, int H) {
  return lstm_persist_ok(dtype, T, N, H) ? persist_ws_bytes(N) : 0;
}

}  // namespace mr
