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
// on timeout the workgroup records a code in the status word and stops waiting (results garbage, no hang).
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

__device__ __forceinline__ void gran_store(u64* p, unsigned value, unsigned tag) {
  __hip_atomic_store((gu64*)p, ((u64)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gran_load(const u64* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave re-reads its NG granules (stride 256 granules) until every tag matches.  Returns false on timeout.
template <int NG>
__device__ __forceinline__ bool sweep(const u64* g, unsigned tag, unsigned (&v)[NG], unsigned* status, unsigned code) {
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const u64 x = gran_load(g + k * 256);
      v[k] = (unsigned)x;
      ok &= (unsigned)(x >> 32) == tag;
    }
    if (__all(ok)) return true;
    if (spins > SPIN_LIMIT) {
      if ((threadIdx.x & 63) == 0) atomicMax(status, code);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
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
};

__global__ __launch_bounds__(256, 1) void lstm_fwd_persist_kernel(LstmPFwd a) {
  typedef Mma<bf16_t>::Frag Frag;
  __shared__ __attribute__((aligned(16))) bf16_t hbuf[2][PR][PLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int g = blockIdx.x % PG, bg = (blockIdx.x / PG) % a.nbg, dir = blockIdx.x / (PG * a.nbg);
  const int row = bg * PR + l15;          // batch row of this lane's accumulator column
  const bool row_ok = row < a.N;
  constexpr int H = PH;

  // ---- the slice of W_hh this wave multiplies, as MFMA fragments held in registers for the whole sequence
  Frag wf[4][8];
  {
    const bf16_t* wbase = a.whh + (long long)dir * 4 * H * H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = g * 4 * PHS + (wave * 4 + i) * 16 + l15;   // gate column (row of whh)
#pragma unroll
      for (int c = 0; c < 8; ++c) wf[i][c] = *(const Frag*)(wbase + (long long)n * H + c * 32 + lg * 8);
    }
  }
  u64* xch = a.xch + ((long long)dir * a.nbg + bg) * 2 * FWD_GRAN;
  float cst[4] = {0.f, 0.f, 0.f, 0.f};
  bool dead = false;

  for (int s = 0; s < a.T; ++s) {
    const int t = dir == 0 ? s : a.T - 1 - s;
    const long long r = (long long)t * a.N + row;
    // operands of the gate math that do not depend on the recurrent term: in flight during the exchange
    f32x4 xg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = g * PHS + (wave * 4 + i) * 4 + lg;
      xg[i] = row_ok ? load4(a.xproj + r * 8 * H + dir * 4 * H + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      // ---- all-gather h_{prev}: 8 granules per thread, tag s, slot (s-1)&1
      unsigned v[8];
      const u64* src = xch + ((s - 1) & 1) * FWD_GRAN + tid;
      if (!dead) dead = !sweep<8>(src, (unsigned)s, v, a.status, 1u);
      bf16_t* hb = &hbuf[s & 1][0][0];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int idx = q * 256 + tid;
        const int ln = idx & 63, p = (idx >> 6) & 1, w2 = (idx >> 7) & 3, g2 = idx >> 9;
        const int m = ln & 15, j0 = g2 * PHS + (w2 * 4 + 2 * p) * 4 + (ln >> 4);
        *(unsigned short*)(hb + m * PLD + j0) = (unsigned short)(v[q] & 0xffffu);
        *(unsigned short*)(hb + m * PLD + j0 + 4) = (unsigned short)(v[q] >> 16);
      }
      __syncthreads();
      Frag hf[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) hf[c] = *(const Frag*)(hb + l15 * PLD + c * 32 + lg * 8);
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<bf16_t>::run(acc[i], wf[i][c], hf[c]);
    }
    // ---- gate math (lane owns the 4 gates of (row, unit) for 4 units), state in registers
    float hv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = g * PHS + (wave * 4 + i) * 4 + lg;
      const float ig = fsig(acc[i][0] + xg[i][0]);
      const float fg = fsig(acc[i][1] + xg[i][1]);
      const float gg = ftanh(acc[i][2] + xg[i][2]);
      const float og = fsig(acc[i][3] + xg[i][3]);
      const float c = fg * cst[i] + ig * gg;
      cst[i] = c;
      hv[i] = og * ftanh(c);
      if (row_ok) {
        a.cbuf[r * 2 * H + dir * H + j] = c;
        a.out[r * 2 * H + dir * H + j] = (bf16_t)hv[i];
        store4(a.gates + r * 8 * H + dir * 4 * H + 4 * j, f32x4{ig, fg, gg, og});
      }
    }
    if (s + 1 < a.T) {
      u64* dst = xch + (s & 1) * FWD_GRAN + ((g * 4 + wave) * 2) * 64 + lane;
      gran_store(dst, pack_bf16(hv[0], hv[1]), (unsigned)(s + 1));
      gran_store(dst + 64, pack_bf16(hv[2], hv[3]), (unsigned)(s + 1));
    }
  }
}

struct LstmPBwd {
  const bf16_t* dout;   // [T, N, 2H]
  const bf16_t* whhT;   // [2][H][4H]  rows = hidden unit, K = gate-interleaved column
  const float* cbuf;    // [T, N, 2H]
  bf16_t* gates;        // [T, N, 8H]  post-activation gates in, pre-activation gradients out (in place)
  u64* xch;             // [2 dirs][nbg][2 slots][BWD_GRAN]
  unsigned* status;
  int T, N, nbg;
  float* dbg;           // debug only (null in production): recurrent term dh_rec [T, N, 2H]
};

__global__ __launch_bounds__(256, 1) void lstm_bwd_persist_kernel(LstmPBwd a) {
  typedef Mma<bf16_t>::Frag Frag;
  __shared__ __attribute__((aligned(16))) bf16_t abuf[PR][PLD];   // own dgates [16 rows][256 own gate columns]
  __shared__ __attribute__((aligned(16))) float obuf[4][4][64];   // own partial dh: [tile][e][lane]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int g = blockIdx.x % PG, bg = (blockIdx.x / PG) % a.nbg, dir = blockIdx.x / (PG * a.nbg);
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
  u64* xch = a.xch + ((long long)dir * a.nbg + bg) * 2 * BWD_GRAN;
  // epilogue ownership: thread (wave i', lane) <-> row l15, local units u0..u0+3 of this slice
  const int u0 = wave * 16 + lg * 4;
  const int j0 = g * PHS + u0;               // global hidden unit of e = 0
  float dcar[4] = {0.f, 0.f, 0.f, 0.f};
  bool dead = false;

  for (int s = 0; s < a.T; ++s) {
    // backward visits the steps in the reverse of the forward order of that direction
    const int t = dir == 0 ? a.T - 1 - s : s;
    const int tp = dir == 0 ? t - 1 : t + 1;     // previous step in forward order (c_prev)
    const bool has_prev = dir == 0 ? t > 0 : t < a.T - 1;
    const long long r = (long long)t * a.N + row;
    f32x4 up = {0.f, 0.f, 0.f, 0.f}, ct = up, cp = up, gq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) gq[e] = up;
    if (row_ok) {
      up = load4(a.dout + r * 2 * H + dir * H + j0);
      ct = *(const f32x4*)(a.cbuf + r * 2 * H + dir * H + j0);
      if (has_prev) cp = *(const f32x4*)(a.cbuf + ((long long)tp * a.N + row) * 2 * H + dir * H + j0);
#pragma unroll
      for (int e = 0; e < 4; ++e) gq[e] = load4(a.gates + r * 8 * H + dir * 4 * H + 4 * (j0 + e));
    }
    f32x4 dh = up;
    if (s > 0) {
      // ---- reduce-scatter: 3 foreign partial sums (tag s, slot (s-1)&1) + the own one from LDS
      unsigned v[12];
      const u64* base = xch + ((s - 1) & 1) * BWD_GRAN + (long long)g * (PG - 1) * BWD_GRAN_SLAB + wave * 256 + lane;
      // granule (src k, e) sits at base + k*BWD_GRAN_SLAB + e*64: fold into the stride-256 sweep by 3 sweeps of 4
      bool ok = true;
      if (!dead) {
        for (unsigned spins = 0;; ++spins) {
          ok = true;
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const u64 x = gran_load(base + k * BWD_GRAN_SLAB + e * 64);
              v[k * 4 + e] = (unsigned)x;
              ok &= (unsigned)(x >> 32) == (unsigned)s;
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
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sum = obuf[wave][e][lane];
#pragma unroll
        for (int k = 0; k < 3; ++k) sum += __uint_as_float(v[k * 4 + e]);
        dh[e] += sum;
        if (a.dbg && row_ok) {   // debug dump: [0] total, [1] own partial, [2..4] foreign partials in slab order
          const long long plane = (long long)a.T * a.N * 2 * H, at = r * 2 * H + dir * H + j0 + e;
          a.dbg[at] = sum;
          a.dbg[plane + at] = obuf[wave][e][lane];
#pragma unroll
          for (int k = 0; k < 3; ++k) a.dbg[(2 + k) * plane + at] = __uint_as_float(v[k * 4 + e]);
        }
      }
    }
    // ---- gate algebra (EpiLstmBwd of lstm.hip with dc carried in registers)
    bf16_t* arow = &abuf[l15][4 * u0];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ig = gq[e][0], fg = gq[e][1], gg = gq[e][2], og = gq[e][3];
      const float tc = ftanh(ct[e]);
      const float dcv = dh[e] * og * (1.f - tc * tc) + dcar[e];
      f32x4 d;
      d[0] = dcv * gg * ig * (1.f - ig);
      d[1] = dcv * cp[e] * fg * (1.f - fg);
      d[2] = dcv * ig * (1.f - gg * gg);
      d[3] = dh[e] * tc * og * (1.f - og);
      dcar[e] = dcv * fg;
      if (row_ok) store4(a.gates + r * 8 * H + dir * 4 * H + 4 * (j0 + e), d);
      store4(arow + 4 * e, row_ok ? d : f32x4{0.f, 0.f, 0.f, 0.f});
    }
    if (s + 1 == a.T) break;
    __syncthreads();   // abuf complete; every thread has consumed obuf of the previous step
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
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) obuf[i][e][lane] = acc[i][e];
    } else {
      // (__builtin_bit_cast on an ext-vector ELEMENT reads element 0 for every index with this compiler: convert
      // through a float rvalue instead)
      const int srcidx = g < wave ? g : g - 1;
      u64* dst = xch + (s & 1) * BWD_GRAN + ((long long)wave * (PG - 1) + srcidx) * BWD_GRAN_SLAB + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          gran_store(dst + i * 256 + e * 64, __float_as_uint((float)acc[i][e]), (unsigned)(s + 1));
    }
    __syncthreads();   // obuf visible; all fragment reads of abuf done before the next step overwrites it
  }
}

static long long persist_ws_bytes(int N) {
  const long long nbg = cdiv(N, PR);
  const long long fwd = 2 * nbg * 2 * FWD_GRAN * 8, bwd = 2 * nbg * 2 * (long long)BWD_GRAN * 8;
  return (fwd > bwd ? fwd : bwd) + 256;
}

// The persistent kernels need all workgroups of a batch group co-resident: keep the grid within the chip.
bool lstm_persist_ok(int dtype, int T, int N, int H) {
  return dtype == MR_BF16 && H == PH && T >= 1 && T < (1 << 30) && 2 * cdiv(N, PR) * PG <= 512;
}

int lstm_fwd_persist(const void* xproj, const void* whh, void* out, float* cbuf, void* gates, int T, int N,
                     void* ws, long long ws_bytes, hipStream_t stream) {
  MR_CHECK_ARG(ws_bytes >= persist_ws_bytes(N), "mr_lstm_fwd: workspace too small (%lld < %lld)", ws_bytes,
               persist_ws_bytes(N));
  const int nbg = cdiv(N, PR);
  const long long xbytes = persist_ws_bytes(N) - 256;
  if (hipMemsetAsync(ws, 0, (size_t)persist_ws_bytes(N), stream) != hipSuccess) {
    set_error("mr_lstm_fwd: memset of the exchange buffer failed");
    return MR_ERR_LAUNCH;
  }
  LstmPFwd a{(const bf16_t*)xproj, (const bf16_t*)whh, (bf16_t*)out, cbuf, (bf16_t*)gates, (u64*)ws,
             (unsigned*)((char*)ws + xbytes), T, N, nbg};
  hipLaunchKernelGGL(lstm_fwd_persist_kernel, dim3(2 * nbg * PG), dim3(256), 0, stream, a);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

static float* g_lstm_bwd_dbg = nullptr;
void lstm_set_bwd_debug(float* p) { g_lstm_bwd_dbg = p; }

int lstm_bwd_persist(const void* dout, const void* whhT, const float* cbuf, void* gates, int T, int N, void* ws,
                     long long ws_bytes, hipStream_t stream) {
  MR_CHECK_ARG(ws_bytes >= persist_ws_bytes(N), "mr_lstm_bwd: workspace too small (%lld < %lld)", ws_bytes,
               persist_ws_bytes(N));
  const int nbg = cdiv(N, PR);
  const long long xbytes = persist_ws_bytes(N) - 256;
  if (hipMemsetAsync(ws, 0, (size_t)persist_ws_bytes(N), stream) != hipSuccess) {
    set_error("mr_lstm_bwd: memset of the exchange buffer failed");
    return MR_ERR_LAUNCH;
  }
  LstmPBwd a{(const bf16_t*)dout, (const bf16_t*)whhT, cbuf, (bf16_t*)gates, (u64*)ws,
             (unsigned*)((char*)ws + xbytes), T, N, nbg, g_lstm_bwd_dbg};
  hipLaunchKernelGGL(lstm_bwd_persist_kernel, dim3(2 * nbg * PG), dim3(256), 0, stream, a);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

long long lstm_persist_ws(int dtype, int T, int N, int H) {
  return lstm_persist_ok(dtype, T, N, H) ? persist_ws_bytes(N) : 0;
}

}  // namespace mr
