// The Bahdanau-attention GRU decode loop as persistent kernels for gfx950: ONE launch for all S steps of the forward pass and one
// for the backward pass, instead of three dependent launches per step and direction (reference: the python loop of
// decoders/attention_decoder.py:84-118 around AttentionRNNCell, :146-231).
//
// Why: a step at the published shape (N = 32 samples, T = 64 positions, H = 512, Ep = 552) is
//   [h W_cat^T]  ->  [energies / softmax / context]  ->  [context W_ic^T + GRU cell]
// = 4.2 + 6.8 + 7.3 us of launch-latency-sized kernels plus the gaps between them (20.4 us per step replayed from a graph; 27.4 us
// for the backward), 32 times, on 16-32 of 256 CUs, each launch re-reading its weights (2 MB + 1.7 MB) from L2.
//
// Decomposition (bf16, H = 512).  Samples are independent through the recurrence, hidden units are not:
//   * the batch is cut into groups of R rows (R = 4 up to 32 samples, else 8: the bytes a workgroup gathers per hand-off scale
//     with R); groups never talk to each other;
//   * a group is 32 workgroups ("slices"); slice g owns the hidden units [16g, 16g+16): their column of the attention hidden
//     projection, their r / z / n rows of W_hh (4 x 16 rows of W_cat, one MFMA tile per wave) and of W_ic (3 tiles).  All of it
//     -- 64 + 54 KB -- is loaded ONCE into the waves' VGPRs and stays there for all S steps; the slice's 16 columns of eproj
//     stay in LDS;
//   * slice g also serves sample g / (32/R) of the group: it holds a 1/(32/R) share of that sample's encoder channels in LDS;
//   * 32 slices are the 32 CUs of one XCD: the block map puts a group on one XCD (decode_roles), the placement is VERIFIED per
//     launch (group_on_one_xcd) and only then are granules published with plain stores that stay in that XCD's L2.
// One forward step = three all-to-all hand-offs inside the group:
//   1. slice g computes hproj / gh of its units (MFMA, h from LDS) and, for every (sample, position), the PARTIAL energy
//      sum_{j in own units} v_j tanh(hproj_j + eproj_j); every slice REDUCES the partials of its sample over the 32 producers
//      (16 KB of granules in), softmax, and the context of its channel share = sum_t w_t enc_t from LDS;
//   2. the R contexts are ALL-GATHERED (R x Ep bf16) into every slice's LDS; GEMM with the W_ic tiles; GRU cell of the
//      slice's units (one thread per (sample, unit));
//   3. h' is ALL-GATHERED (R x 512 bf16) into every slice's LDS: the next step's MFMA operand.
//   With arg-max feedback (flags[s] == 0: attention_decoder.py:107-110) wave 3 also scores the slice's 8 classes of the output
//   layer on the h' it holds; the best (logit, class) of every (sample, slice) rides in the energies sweep of the next step.
// The backward kernel (second half of the file) runs the steps in reverse with three reduce-scatters of f32 partial sums.
// Protocol: lstm_persist.hip's -- a granule is one naturally aligned 8-byte {value, tag} written by one store and polled by sc1
// (agent scope, L1-bypassing) loads; tag = step + 1, never 0; the exchange buffer is zeroed ahead of the launch; two slots
// alternate (every hand-off is all-to-all inside the group, so no producer can run two steps ahead of a consumer).
// Every spin is bounded: on timeout the workgroup records a code in the status word, stops waiting and POISONS h' with NaN.
//
// Results: the buffers the per-step path saves for the backward (H_all, HC_all, W_att, CTX_all, SAVE_all), rounded at the same
// points (hproj / gh / context / h to bf16, the context part of the input gates kept in f32), so either backward can follow.
#include "common.h"
#include "igemm_core.h"
#include "../../include/megreader_hip.h"

namespace mr {

namespace {

constexpr int DH = 512;          // hidden size
constexpr int DG = 32;           // slices per batch group
constexpr int DU = DH / DG;      // hidden units per slice (16: one MFMA tile)
constexpr int DT = 64;           // positions (max)
constexpr int DEPMAX = 576;      // encoder channels (max, multiple of 32)
constexpr int HLD = DH + 8;      // LDS row stride of h (elements): conflict-free 16-byte fragment reads
constexpr int CLD = DEPMAX + 8;  // LDS row stride of the contexts
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr unsigned TIMING_MAGIC = 0x54494D45u;

typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int AUX_SC1 = 16;

// exchange layout of one batch group of R rows (bytes)
template <int R>
struct Xch {
  static constexpr unsigned XS_SLOT = DG * R * DT * 8;           // partial energies: [producer][sample][position] {f32, tag}
  static constexpr unsigned XC_SLOT = R * (DEPMAX / 4) * 16;     // contexts: [sample][4-channel unit] {bf16x2, tag} x 2
  static constexpr unsigned XH_SLOT = R * (DH / 4) * 16;         // hidden:   [sample][4-unit unit]
  static constexpr unsigned XL_SLOT = R * DG * 16;               // best class: [sample][slice] {logit, tag, class, tag}
  static constexpr unsigned XS_OFF = 0, XC_OFF = 2 * XS_SLOT, XH_OFF = XC_OFF + 2 * XC_SLOT, XL_OFF = XH_OFF + 2 * XH_SLOT;
  static constexpr unsigned GROUP = XL_OFF + 2 * XL_SLOT;
  static constexpr int CNT_C = (R * (DEPMAX / 4) + 255) / 256;   // 16-byte pairs per thread of the context gather
  static constexpr int CNT_H = R * (DH / 4) / 256;               // ... of the hidden gather
};

__device__ __forceinline__ void gran2_store(rsrc_t r, unsigned byte_off, unsigned v0, unsigned v1, unsigned tag) {
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, tag, v1, tag}, r, (int)byte_off, 0, AUX_SC1);
}
// `local` (uniform): all 32 slices of the batch group were found on ONE XCD (group_on_one_xcd below).  A plain store then KEEPS the
// line in that XCD's L2 and the siblings' sc1 (L1-bypassing) polls hit it there; an sc1 store drops the line from L2 and every
// poll pays the fabric round trip (MI355X_MICROARCH.md, "stores of each flavour"; lstm_persist.hip does the same with 4 slices).
__device__ __forceinline__ void gran2_publish(rsrc_t r, unsigned byte_off, unsigned v0, unsigned v1, unsigned tag, bool local) {
  if (local)
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, tag, v1, tag}, r, (int)byte_off, 0, 0);
  else
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, tag, v1, tag}, r, (int)byte_off, 0, AUX_SC1);
}
__device__ __forceinline__ u32x4 gran2_load(rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX_SC1);
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16_t x = (bf16_t)a, y = (bf16_t)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ int wave_of_thread() { return (int)(threadIdx.x >> 6); }
// sigmoid on the hardware exp2 / rcp units (the GRU cell sits on the chain: the library forms cost ~0.15 us per step; absolute
// error ~2e-7, h is rounded to bf16 right after -- lstm_persist.hip does the same)
__device__ __forceinline__ float dec_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
// the energies' tanh: the form of attention.hip (att_tanh)
__device__ __forceinline__ float dec_tanh(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.885390081777927f));
}

// Cross-lane sums on the DPP path (one VALU op a level; __shfl_xor is a ds_bpermute round trip a level and the softmax of a step
// sits on the chain 12 levels deep).  After the two quad permutations every quad is uniform, so the (half-)row mirrors act as
// xor 4 / xor 8.  All lanes of the wave must be executing.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int LANES>   // sum over aligned groups of LANES (2, 4, 8, 16) adjacent lanes, result in every lane of the group
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_mov<0xB1>(v);                            // quad_perm [1,0,3,2]
  if constexpr (LANES >= 4) v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
  if constexpr (LANES >= 8) v += dpp_mov<0x141>(v); // row_half_mirror
  if constexpr (LANES >= 16) v += dpp_mov<0x140>(v);// row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = group_sum<16>(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Gather CNT 16-byte granule pairs (bit k of `want`: this thread needs pair k at byte offset base + off[k]; the other offsets must
// still point into the workspace); a pair is accepted when both of its tags equal `tag`.  Every sweep issues ALL its loads before
// it looks at any of them: with a load inside the per-pair branch the compiler put `s_waitcnt vmcnt(0)` behind each one (seen in
// the ISA) and a 5-pair sweep paid five L2 round trips in a row.  The wave leaves together.  Returns false on timeout.
template <int CNT>
__device__ __forceinline__ bool gather_pairs(rsrc_t rx, const unsigned (&off)[CNT], unsigned base, unsigned want, unsigned tag,
                                             u32x4 (&v)[CNT]) {
  unsigned need = want;
  for (unsigned spins = 0;; ++spins) {
    u32x4 t[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) t[k] = gran2_load(rx, base + off[k]);
    __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise sinks the last load behind the first wait)
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
      if (((need >> k) & 1u) && t[k][1] == tag && t[k][3] == tag) {
        v[k] = t[k];
        need &= ~(1u << k);
      }
    }
    if (__all(need == 0)) return true;
    if (spins > SPIN_LIMIT) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}


constexpr unsigned HELLO_TAG = 0x48454C4Fu;
constexpr unsigned HELLO_BYTES = 8 * DG * 16;     // [8 groups][32 slices] granule pairs, between the exchange slots and the status

// Workgroup -> (slice g, batch group bg).  xmap: the grid is 8 x 32 workgroups and hardware workgroup b is dispatched to XCD
// b % 8 (round-robin), so group bg takes the block indices congruent to bg mod 8: its 32 slices fill the 32 CUs of ONE XCD (one
// workgroup per CU: 460+ registers a lane).  Returns false for the workgroups of groups that do not exist.
__device__ __forceinline__ bool decode_roles(int xmap, int nbg, int& g, int& bg) {
  const int b = blockIdx.x;
  if (xmap) {
    bg = b & 7;
    g = b >> 3;
    return bg < nbg;
  }
  g = b % DG;
  bg = b / DG;
  return true;
}

// Are the 32 slices of this batch group on one XCD?  The placement above is a dispatch-order ASSUMPTION, so it is verified: every
// workgroup publishes its HW_REG_XCC_ID (sc1 store: visible anywhere) and reads its 31 siblings'.  Only if all agree does this
// workgroup publish with plain stores.  A sibling that does not answer within the spin bound counts as "elsewhere" (sc1 stores
// are always correct).  `sh` is one LDS word; status word 1 counts the workgroups that answered yes.
__device__ __forceinline__ bool group_on_one_xcd(rsrc_t rx, unsigned hello_base, int g, int xmap, int* sh, unsigned* status) {
  if (!xmap) return false;
  unsigned me;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(me));
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) gran2_store(rx, hello_base + (unsigned)(g * 16), me, me, HELLO_TAG);
  if (tid < 64) {
    bool same = true;
    if (lane < DG && lane != g) {
      same = false;
      for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
        const u32x4 v = gran2_load(rx, hello_base + (unsigned)(lane * 16));
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

struct DecP {
  const bf16_t* cat_w;     // [4H][H]: rows [0,H) attention hidden projection, [H,4H) W_hh (r, z, n)
  const float* cat_b;      // [4H] or null
  const bf16_t* ic_w;      // [3H][ldic]: context columns of W_ih
  long long ldic;
  const bf16_t* G;         // word table [classes][ldG >= 3H] (W_ih word part + b_ih, gathered by idx)
  long long ldG;
  long long* idx;          // [S][N] word fed to each step (flags != null: rows of arg-max steps are WRITTEN by the kernel)
  const int* flags;        // [S] or null (= every step is fed the word in idx).  flags[s] == 0: step s + 1 is fed the arg-max of
                           // step s's output layer (attention_decoder.py:107-110) instead of idx[s + 1]
  const bf16_t* out_w;     // [C][H] output layer (flags != null), C <= 256: slice g scores the classes [8g, 8g + 8)
  const float* out_b;      // [C] or null
  int C;
  const bf16_t* eproj;     // [N][T][H]
  const bf16_t* enc;       // [N][T][Ep]
  const float* v;          // [H]
  bf16_t* H_all;           // [S+1][N][H]   ([0] is the initial state: read)
  bf16_t* HC_all;          // [S][N][4H]
  float* W_att;            // [S][N][T]
  bf16_t* CTX_all;         // [S][N][Ep]
  float* SAVE_all;         // [S][N][3H]  r, z, n
  u64* xch;
  unsigned* status;
  int S, N, T, Ep, nbg;
  int xmap;                // block -> role map that puts a group's 32 slices on one XCD (decode_roles)
  unsigned hello_off;      // byte offset (from xch) of the XCC-id exchange
};

// LDS carve-up (bytes), shared by the kernel and the launcher
template <int R>
struct Lds {
  static constexpr int HBUF = 0;                                  // bf16 [16][HLD]   (rows >= R stay zero)
  static constexpr int EP = HBUF + 16 * HLD * 2;                  // uint4 [2 halves][R * 64 (sample, position)]: 8 units each
  static constexpr int HC = EP + 2 * R * DT * 16;                 // f32 [4 tiles][16 rows][16 units]
  static constexpr int GI = HC + 4 * 16 * DU * 4;                 // f32 [3 gates][16 rows][16 units]
  static constexpr int SC = GI + 3 * 16 * DU * 4;                 // f32 [8][64] partial energies, [64] weights
  static constexpr int DEAD = SC + 9 * 64 * 4;                    // int [4]
  static constexpr int WORD = DEAD + 16;                          // int [16] fed word of each row (arg-max steps)
  static constexpr int CTX = WORD + 64;                           // bf16 [16][CLD]   (rows >= R, columns >= Ep stay zero)
  static constexpr int ENC = CTX + 16 * CLD * 2;                  // bf16 [T][4 * upp]: this slice's channel share of its sample
  static size_t bytes(int T, int Ep) {
    const int upp = cdiv(Ep / 4, DG / R);
    return (size_t)ENC + (size_t)T * upp * 8;
  }
};

}  // namespace

template <int R>
__global__ __launch_bounds__(256, 1) void decode_fwd_persist_kernel(DecP a) {
  typedef Mma<bf16_t>::Frag Frag;
  typedef Xch<R> X;
  typedef Lds<R> L;
  constexpr int SPS = DG / R;        // slices serving one sample
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* hbuf = (bf16_t*)(smem + L::HBUF);
  uint4* sEp = (uint4*)(smem + L::EP);
  float* sHC = (float*)(smem + L::HC);
  float* sGI = (float*)(smem + L::GI);
  float* sSc = (float*)(smem + L::SC);
  int* sDead = (int*)(smem + L::DEAD);
  int* sWord = (int*)(smem + L::WORD);
  bf16_t* sCtx = (bf16_t*)(smem + L::CTX);
  bf16_t* sEnc = (bf16_t*)(smem + L::ENC);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  int g, bg;
  if (!decode_roles(a.xmap, a.nbg, g, bg)) return;
  const int T = a.T, Ep = a.Ep, N = a.N, S = a.S;
  const int nu = Ep >> 2;                       // 4-channel units per context row
  const int row_f = bg * R + l15;               // batch row of this lane's MFMA output column
  const bool row_f_ok = l15 < R && row_f < N;
  // the sample this slice reduces the energies of, and its share of that sample's context channels
  const int nloc = g / SPS, cpart = g % SPS, row_o = bg * R + nloc;
  const bool row_o_ok = row_o < N;
  const int upp = (nu + SPS - 1) / SPS;         // 4-channel units per share
  const int u0 = cpart * upp;                   // first unit of this slice's share
  const int nown = max(0, min(upp, nu - u0));   // units of the share that exist
  // context sum: TG adjacent lanes share a unit, each takes the positions t = ctg (mod TG); upp * TG <= 144 threads
  constexpr int TG = 32 / R;
  const int cu = tid / TG, ctg = tid % TG;      // this thread's (unit, position group)
  const bool ctx_thread = cu < nown;

  // ---- weights: VGPR-resident MFMA fragments.  wave w: W_cat tile w (w = 0: hproj, 1..3: gh r/z/n); waves 0..2: W_ic gate w
  Frag wcat[16], wic[18];
  {
    const int rc = (wave == 0 ? 0 : DH + (wave - 1) * DH) + g * DU + l15;
    const bf16_t* p = a.cat_w + (long long)rc * DH + lg * 8;
#pragma unroll
    for (int c = 0; c < 16; ++c) wcat[c] = *(const Frag*)(p + c * 32);
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const bf16_t* q = a.ic_w + (long long)((wave < 3 ? wave : 0) * DH + g * DU + l15) * a.ldic + lg * 8;
    // wave 3 has no W_ic tile: with arg-max feedback its fragments hold the slice's 8 rows of the output layer instead
    const int cls = g * 8 + l15;
    const bool out_row = a.flags != nullptr && wave == 3 && l15 < 8 && cls < a.C;
    const bf16_t* qo = a.out_w + (long long)(out_row ? cls : 0) * DH + lg * 8;
#pragma unroll
    for (int c = 0; c < 18; ++c) {
      uint4 t = z4;
      if (wave < 3 && c * 32 + lg * 8 < Ep) t = *(const uint4*)(q + c * 32);
      if (out_row && c < 16) t = *(const uint4*)(qo + c * 32);
      wic[c] = *(const Frag*)&t;
    }
  }
  const bool coin = a.flags != nullptr;
  f32x4 obias = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // classes that do not exist never win
  if (coin && wave == 3 && lg < 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = g * 8 + lg * 4 + q;
      if (c < a.C) obias[q] = a.out_b ? a.out_b[c] : 0.f;
    }
  }
  const int colbase = (wave == 0 ? 0 : DH + (wave - 1) * DH) + g * DU + lg * 4;   // first of this lane's 4 W_cat columns
  f32x4 cbias = {0.f, 0.f, 0.f, 0.f};
  if (a.cat_b) cbias = *(const f32x4*)(a.cat_b + colbase);
  float vv[DU];
#pragma unroll
  for (int j = 0; j < DU; j += 4) {
    const f32x4 t = *(const f32x4*)(a.v + g * DU + j);
    vv[j] = t[0]; vv[j + 1] = t[1]; vv[j + 2] = t[2]; vv[j + 3] = t[3];
  }
  // ---- LDS residents: the slice's eproj columns, its encoder share, h_0, zero padding of the context tile
  for (int p = tid; p < R * DT; p += 256) {
    const int n = p >> 6, t = p & 63, r = bg * R + n;
    uint4 e0 = make_uint4(0, 0, 0, 0), e1 = e0;
    if (r < N && t < T) {
      const bf16_t* ep = a.eproj + ((long long)r * T + t) * DH + g * DU;
      e0 = *(const uint4*)ep;
      e1 = *(const uint4*)(ep + 8);
    }
    sEp[p] = e0;
    sEp[R * DT + p] = e1;
  }
  for (int i = tid; i < T * upp; i += 256) {
    const int t = i / upp, u = i - t * upp;
    uint2 e = make_uint2(0, 0);
    if (row_o_ok && u < nown) e = *(const uint2*)(a.enc + ((long long)row_o * T + t) * Ep + (u0 + u) * 4);
    *(uint2*)(sEnc + (long long)i * 4) = e;
  }
  for (int i = tid; i < 16 * HLD / 8; i += 256) {
    const int r = i / (HLD / 8), c = (i - r * (HLD / 8)) * 8, rr = bg * R + r;
    uint4 h0 = make_uint4(0, 0, 0, 0);
    if (r < R && rr < N && c < DH) h0 = *(const uint4*)(a.H_all + (long long)rr * DH + c);
    *(uint4*)(hbuf + r * HLD + c) = h0;
  }
  for (int i = tid; i < 16 * CLD / 8; i += 256) ((uint4*)sCtx)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 4) sDead[tid] = 0;

  const rsrc_t rx = make_rsrc(a.xch);
  const unsigned xg = (unsigned)bg * X::GROUP;
  const bool local = group_on_one_xcd(rx, a.hello_off + (unsigned)(bg * DG * 16), g, a.xmap, &sDead[3], a.status);
  // per-thread constants of the gathers
  unsigned offC[X::CNT_C], ldsC[X::CNT_C], wantC = 0;      // contexts: R * nu pairs
#pragma unroll
  for (int k = 0; k < X::CNT_C; ++k) {
    const int u = tid + 256 * k;
    offC[k] = (unsigned)u * 16u;
    const int r = u / nu, c = u - r * nu;
    ldsC[k] = (unsigned)(r * CLD + c * 4);
    if (u < R * nu) wantC |= 1u << k;
    else offC[k] = 0u;                       // (every offset is loaded from)
  }
  unsigned offH[X::CNT_H];                                 // hidden: R * 128 pairs, linear
#pragma unroll
  for (int k = 0; k < X::CNT_H; ++k) offH[k] = (unsigned)(tid + 256 * k) * 16u;
  unsigned offS[4];                                        // the partial energies of this slice's sample from 32 producers
  {
    const int tp = tid & 31, pg = tid >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) offS[k] = (unsigned)(((pg + 8 * k) * R * DT + nloc * DT + tp * 2) * 8);
  }
  // GRU ownership: thread <-> (sample gm, unit gu of the slice)
  const int gm = tid >> 4, gu = tid & 15, row_g = bg * R + gm, jg = g * DU + gu;
  const bool gru_thread = gm < R;
  const bool row_g_ok = gru_thread && row_g < N;
  bool dead = false;
  // the word rows of the GRU's input gates, fetched one step ahead (idx -> row is a dependent pair of loads)
  long long widx = row_g_ok ? a.idx[row_g] : 0;
  unsigned short gw[3] = {0, 0, 0};       // raw bf16: converted where they are used, so the loads are waited for there
  if (row_g_ok) {
    const unsigned short* gp = (const unsigned short*)(a.G + widx * a.ldG + jg);
    gw[0] = gp[0]; gw[1] = gp[DH]; gw[2] = gp[2 * DH];
  }
  __syncthreads();
  // phase clock (tools/microbench_decode.py): thread 0 of slices 0 and 1 of group 0 adds the 100 MHz wall clock spent in each
  // phase of a step into status words [8 + 16 g + phase]; only when the caller set status word 2 (the product never does)
  const bool timing = tid == 0 && bg == 0 && g < 2 && a.status[2] == TIMING_MAGIC;
  unsigned long long tprev = timing ? wall_clock64() : 0ull;
  unsigned tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DEC_TICK(i)                                   \
  if (timing) {                                       \
    const unsigned long long tn_ = wall_clock64();    \
    tacc[i] += (unsigned)(tn_ - tprev);               \
    tprev = tn_;                                      \
  }

  int flag_prev = 1;             // flags[s - 1] while step s runs
  for (int s = 0; s < S; ++s) {
    const unsigned tag = (unsigned)(s + 1);
    const unsigned slot = (unsigned)(s & 1);
    // arg-max feedback: is the word of THIS step the arg-max of the previous step's output layer?  (uniform)
    // (the coin of step s is fetched during step s: read here it was a global round trip at the top of every step)
    const bool fed_argmax = coin && s > 0 && flag_prev == 0;
    if (coin) flag_prev = a.flags[s];
    if (coin) {
      if (row_g_ok) widx = a.idx[(long long)s * N + row_g];                        // the given word (ignored on arg-max steps)
    } else if (row_g_ok && s + 1 < S) {
      widx = a.idx[(long long)(s + 1) * N + row_g];                                // used after the context hand-off
    }
    // ---- 1. stacked hidden projection of the slice's units: [16 rows] x [this wave's 16 columns of W_cat]
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, lo0 = acc0, lo1 = acc0;
      const bool do_logits = fed_argmax && wave == 3;      // h in LDS is h' of the previous step: its output layer, 8 classes
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
        const Frag h0 = *(const Frag*)(hbuf + l15 * HLD + c * 32 + lg * 8);
        const Frag h1 = *(const Frag*)(hbuf + l15 * HLD + c * 32 + 32 + lg * 8);
        Mma<bf16_t>::run(acc0, wcat[c], h0);      // D[column][row]: this lane = 4 consecutive columns of row l15
        Mma<bf16_t>::run(acc1, wcat[c + 1], h1);
        if (do_logits) {
          Mma<bf16_t>::run(lo0, wic[c], h0);
          Mma<bf16_t>::run(lo1, wic[c + 1], h1);
        }
      }
      if (do_logits) {
        // best class of this slice for sample l15 (first index wins ties, as mr_out_nll_fwd): lanes lg = 0 / 1 hold 4 classes each
        const f32x4 lgt = (lo0 + lo1) + obias;
        float bv = lgt[0];
        int bc = g * 8 + lg * 4;
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if (lgt[q] > bv) { bv = lgt[q]; bc = g * 8 + lg * 4 + q; }
        const float ov = __shfl_xor(bv, 16, 64);
        const int oc = __shfl_xor(bc, 16, 64);
        if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
        if (lg == 0 && l15 < R)
          gran2_publish(rx, xg + X::XL_OFF + slot * X::XL_SLOT + (unsigned)((l15 * DG + g) * 16), __float_as_uint(bv), (unsigned)bc,
                      tag, local);
      }
      const f32x4 acc = acc0 + acc1;
      const unsigned p01 = pack_bf16(acc[0] + cbias[0], acc[1] + cbias[1]);
      const unsigned p23 = pack_bf16(acc[2] + cbias[2], acc[3] + cbias[3]);
      *(f32x4*)(sHC + (wave * 16 + l15) * DU + lg * 4) = f32x4{bf16_lo(p01), bf16_hi(p01), bf16_lo(p23), bf16_hi(p23)};
      if (row_f_ok) *(uint2*)(a.HC_all + ((long long)s * N + row_f) * 4 * DH + colbase) = make_uint2(p01, p23);
    }
    lds_barrier();
    DEC_TICK(0)
    // ---- partial energies of (sample w + 4i, position lane) over the slice's 16 units -> the slices of that sample
    {
      const unsigned xs = xg + X::XS_OFF + slot * X::XS_SLOT + (unsigned)(g * R * DT * 8);
#pragma unroll
      for (int i = 0; i < R / 4; ++i) {
        const int n = wave + 4 * i, p = n * DT + lane;
        const uint4 e0 = sEp[p], e1 = sEp[R * DT + p];
        const float* hp = sHC + n * DU;          // tile 0 = hproj
        const unsigned ew[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        float sc0 = 0.f, sc1 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          sc0 += vv[2 * q] * dec_tanh(hp[2 * q] + bf16_lo(ew[q]));
          sc1 += vv[2 * q + 1] * dec_tanh(hp[2 * q + 1] + bf16_hi(ew[q]));
        }
        float sc = sc0 + sc1;
        if (lane >= T) sc = 0.f;
        const float nb = __shfl_down(sc, 1, 64);
        if (!(lane & 1)) gran2_publish(rx, xs + (unsigned)(p * 8), __float_as_uint(sc), __float_as_uint(nb), tag, local);
      }
    }
    DEC_TICK(1)
    // ---- 2. reduce the partials of this slice's sample, softmax, context of its channel share
    {
      u32x4 pv[5];
      unsigned offS5[5] = {offS[0], offS[1], offS[2], offS[3], 0u};
      // arg-max steps: the 32 slices' best classes of every sample ride in the same sweep (thread <-> (sample, slice))
      const unsigned wantS = 0xfu | ((fed_argmax && tid < R * DG) ? 0x10u : 0u);
      offS5[4] = (wantS & 0x10u) ? (X::XL_OFF + slot * X::XL_SLOT + (unsigned)tid * 16u) - (X::XS_OFF + slot * X::XS_SLOT) : offS[0];
      pv[4] = u32x4{0u, 0u, 0u, 0u};
      if (!dead && !gather_pairs<5>(rx, offS5, xg + X::XS_OFF + slot * X::XS_SLOT, wantS, tag, pv)) {
        dead = true;
        if (lane == 0) { atomicMax(a.status, 1u); sDead[0] = 1; }
      }
      DEC_TICK(2)
      if (fed_argmax && tid < R * DG) {         // (whole waves: R * 32 threads)
        // arg-max over the 32 slices of a sample = over a half wave: doubling shifts inside the rows of 16, then the two rows
        float bv = __uint_as_float(pv[4][0]);
        int bc = (int)pv[4][2];
#define DEC_ARGMAX_STEP(CTRL)                                                                                   \
  {                                                                                                             \
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, bv),         \
                                                                           __builtin_bit_cast(int, bv), CTRL, 0xf, 0xf, false)); \
    const int oc = __builtin_amdgcn_update_dpp(bc, bc, CTRL, 0xf, 0xf, false);                                  \
    if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }                                                 \
  }
        DEC_ARGMAX_STEP(0x111) DEC_ARGMAX_STEP(0x112) DEC_ARGMAX_STEP(0x114) DEC_ARGMAX_STEP(0x118)
#undef DEC_ARGMAX_STEP
        const int hb = lane & 32;
        const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 15));
        const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 31));
        const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 47));
        const float v3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 63));
        const int c0 = __builtin_amdgcn_readlane(bc, 15), c1 = __builtin_amdgcn_readlane(bc, 31);
        const int c2 = __builtin_amdgcn_readlane(bc, 47), c3 = __builtin_amdgcn_readlane(bc, 63);
        const float va = hb ? v2 : v0, vb = hb ? v3 : v1;
        const int ca = hb ? c2 : c0, cb = hb ? c3 : c1;
        const int word = (vb > va || (vb == va && cb < ca)) ? cb : ca;
        if ((lane & 31) == 0) {
          const int r = tid >> 5;
          sWord[r] = dead ? 0 : word;
          if (g == 0 && bg * R + r < N) a.idx[(long long)s * N + bg * R + r] = dead ? 0 : word;
        }
      }
      float s0 = 0.f, s1 = 0.f;
      if (!dead) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0 += __uint_as_float(pv[k][0]); s1 += __uint_as_float(pv[k][2]); }
      }
      {
        const int tp = tid & 31, pg = tid >> 5;
        *(float2*)(sSc + pg * 64 + tp * 2) = make_float2(s0, s1);
      }
      lds_barrier();
      if (coin && row_g_ok) {                 // the word rows of this step's input gates: known only now on arg-max steps
        const long long w = fed_argmax ? (long long)sWord[gm] : widx;
        const unsigned short* gp = (const unsigned short*)(a.G + w * a.ldG + jg);
        gw[0] = gp[0]; gw[1] = gp[DH]; gw[2] = gp[2 * DH];
      }
      {
        // every wave runs the 64-wide softmax redundantly (lane t of every wave holds w_t): no barrier before the context sum
        float e = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) e += sSc[k * 64 + lane];
        e = lane < T ? e : -INFINITY;
        const float mx = wave_max_dpp(e);
        const float ex = lane < T ? __builtin_amdgcn_exp2f((e - mx) * 1.4426950408889634f) : 0.f;
        const float sm = wave_sum_dpp(ex);
        const float w = ex * __builtin_amdgcn_rcpf(sm);
        if (wave == 0 && cpart == 0 && lane < T && row_o_ok) a.W_att[((long long)s * N + row_o) * T + lane] = w;
        // this thread's positions of the context sum: t = ctg, ctg + TG, ...  (every lane runs every iteration: a shuffle
        // reads 0 from a lane that is not executing)
        f32x4 cacc = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* er = sEnc + cu * 4;
#pragma unroll
        for (int i = 0; i < DT / TG; ++i) {
          const int t = ctg + i * TG;
          const bool ok = ctx_thread && t < T;
          const uint2 ev = ok ? *(const uint2*)(er + t * upp * 4) : make_uint2(0, 0);
          const float wt = __shfl(w, t, 64);
          cacc[0] += wt * bf16_lo(ev.x); cacc[1] += wt * bf16_hi(ev.x);
          cacc[2] += wt * bf16_lo(ev.y); cacc[3] += wt * bf16_hi(ev.y);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cacc[q] = group_sum<TG>(cacc[q]);
        DEC_TICK(3)
        if (ctx_thread && ctg == 0) {
          const unsigned p01 = pack_bf16(cacc[0], cacc[1]), p23 = pack_bf16(cacc[2], cacc[3]);
          gran2_publish(rx, xg + X::XC_OFF + slot * X::XC_SLOT + (unsigned)((nloc * nu + u0 + cu) * 16), p01, p23, tag, local);
          if (row_o_ok) *(uint2*)(a.CTX_all + ((long long)s * N + row_o) * Ep + (u0 + cu) * 4) = make_uint2(p01, p23);
        }
      }
    }
    DEC_TICK(4)
    // ---- all-gather of the R contexts -> LDS
    {
      u32x4 cv[X::CNT_C];
      if (!dead && !gather_pairs<X::CNT_C>(rx, offC, xg + X::XC_OFF + slot * X::XC_SLOT, wantC, tag, cv)) {
        dead = true;
        if (lane == 0) { atomicMax(a.status, 2u); sDead[1] = 1; }
      }
      if (!dead) {
#pragma unroll
        for (int k = 0; k < X::CNT_C; ++k)
          if ((wantC >> k) & 1u) *(uint2*)(sCtx + ldsC[k]) = make_uint2(cv[k][0], cv[k][2]);
      }
    }
    lds_barrier();
    DEC_TICK(5)
    // ---- context part of the input gates (f32), GRU cell of the slice's units
    if (wave < 3) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int c = 0; c < 18; c += 2) {
        const Frag c0 = *(const Frag*)(sCtx + l15 * CLD + c * 32 + lg * 8);
        const Frag c1 = *(const Frag*)(sCtx + l15 * CLD + c * 32 + 32 + lg * 8);
        Mma<bf16_t>::run(acc0, wic[c], c0);
        Mma<bf16_t>::run(acc1, wic[c + 1], c1);
      }
      *(f32x4*)(sGI + (wave * 16 + l15) * DU + lg * 4) = acc0 + acc1;
    }
    lds_barrier();
    DEC_TICK(6)
    if (gru_thread) {
      const int o = gm * DU + gu;
      const float ir = bf16_lo(gw[0]) + sGI[o], iz = bf16_lo(gw[1]) + sGI[16 * DU + o],
                  in_ = bf16_lo(gw[2]) + sGI[2 * 16 * DU + o];
      const float hr = sHC[16 * DU + o], hz = sHC[2 * 16 * DU + o], hn = sHC[3 * 16 * DU + o];
      const float r = dec_sigmoid(ir + hr), z = dec_sigmoid(iz + hz);
      const float nn_ = dec_tanh(in_ + r * hn);
      const float hp = (float)hbuf[gm * HLD + jg];
      float hnew = (1.f - z) * nn_ + z * hp;
      if (sDead[0] | sDead[1] | sDead[2]) hnew = __builtin_nanf("");
      if (row_g_ok) {
        float* sv = a.SAVE_all + ((long long)s * N + row_g) * 3 * DH + jg;
        sv[0] = r; sv[DH] = z; sv[2 * DH] = nn_;
      }
      if (!row_g_ok) hnew = 0.f;
      if (!coin && row_g_ok && s + 1 < S) {        // the next step's word rows: a whole step to arrive
        const unsigned short* gp = (const unsigned short*)(a.G + widx * a.ldG + jg);
        gw[0] = gp[0]; gw[1] = gp[DH]; gw[2] = gp[2 * DH];
      }
      const float h1 = __shfl_down(hnew, 1, 64), h2 = __shfl_down(hnew, 2, 64), h3 = __shfl_down(hnew, 3, 64);
      if (!(gu & 3)) {
        const unsigned p01 = pack_bf16(hnew, h1), p23 = pack_bf16(h2, h3);
        if (s + 1 < S)
          gran2_publish(rx, xg + X::XH_OFF + slot * X::XH_SLOT + (unsigned)((gm * (DH / 4) + (jg >> 2)) * 16), p01, p23, tag, local);
        if (row_g_ok) *(uint2*)(a.H_all + ((long long)(s + 1) * N + row_g) * DH + jg) = make_uint2(p01, p23);
      }
    }
    DEC_TICK(7)
    if (s + 1 == S) break;
    // ---- 3. all-gather of h' -> LDS
    {
      u32x4 hv[X::CNT_H];
      const bool okh =
          dead || gather_pairs<X::CNT_H>(rx, offH, xg + X::XH_OFF + slot * X::XH_SLOT, (1u << X::CNT_H) - 1u, tag, hv);
      if (!okh) {
        dead = true;
        if (lane == 0) { atomicMax(a.status, 3u); sDead[2] = 1; }
      }
      lds_barrier();        // every GRU thread has read its h
      if (!dead) {
#pragma unroll
        for (int k = 0; k < X::CNT_H; ++k) {
          const int u = tid + 256 * k, r = u >> 7, c = (u & 127) * 4;
          *(uint2*)(hbuf + r * HLD + c) = make_uint2(hv[k][0], hv[k][2]);
        }
      }
    }
    lds_barrier();
    DEC_TICK(8)
  }
  if (timing) {
#pragma unroll
    for (int i = 0; i < 10; ++i) a.status[8 + 16 * g + i] = tacc[i];
  }
#undef DEC_TICK
}

// ------------------------------------------------------------------------------------------------------------------ backward
// Persistent BACKWARD of the decode loop: all S steps (in reverse) in one launch, the same 32 slices per group of R rows.  The
// per-step path is [GEMM(dHC W_cat) + GRU backward] -> [GEMM(dgi W_ic)] -> [attention backward] = 12.9 + 6.6 + 7.4 us a step.
//   Slice g keeps, for its 16 hidden units, the K = 48 rows (r, z, n gate units) of W_ic^T and the K = 64 rows (hproj column +
//   gh r/z/n) of W_cat^T as VGPR-resident MFMA fragments: both GEMMs of a step have their REDUCTION dimension distributed over
//   the slices, so each produces f32 partial sums that are REDUCE-SCATTERED to the slice that consumes them:
//   1. (carried dh_b in a register) + reduce of the 32 partial dh_a of the previous iteration + dh_c -> GRU backward of the own
//      units (dgi -> DGI_all, dgh -> DHC_all);  partial dctx = dgi[own 48] W_ic[own 48, :]            -> edge A (R x Ep f32)
//   2. slice (sample n, channel share c) reduces its share of dctx[n] over the 32 producers (-> DCTX_all) and the partial
//      dw[t] = dctx[share] . enc[n, t, share] from LDS                                                -> edge B (R x 32/R x 64 f32)
//   3. every slice sums the partial dw of every sample, softmax backward de = w (dw - w . dw), then for its 16 units the tanh
//      chain: dhproj (-> DHC_all), deproj and dv accumulated IN REGISTERS over all steps (written once at the end);
//      partial dh_a = dHC[own 64] W_cat[own 64, :]                                                    -> edge C (R x 512 f32)
namespace {
struct DecB {
  const bf16_t* cat_wt;    // [H][4H]: row = hidden unit, K = stacked column
  const bf16_t* ic_wt;     // [Ep][ldict]: row = context channel, K = gate unit (3H)
  long long ldict;
  const bf16_t* eproj;     // [N][T][H]
  const bf16_t* enc;       // [N][T][Ep]
  const float* v;          // [H]
  const bf16_t* H_all;     // [S+1][N][H]
  const bf16_t* HC_all;    // [S][N][4H]
  const float* W_att;      // [S][N][T]
  const float* SAVE_all;   // [S][N][3H]
  const bf16_t* DHO_all;   // [S][N][H]  gradient of h' from the output layer
  const float* ga;         // gradient of the attention weights: ga[n * ldga + s * T + t], or null
  long long ldga;
  bf16_t* DGI_all;         // [S][N][3H]
  bf16_t* DHC_all;         // [S][N][4H]
  bf16_t* DCTX_all;        // [S][N][Ep]
  float* deproj;           // [N][T][H]   written
  float* dv;               // [H]         atomically added to
  bf16_t* denc;            // [N][T][Ep]  gradient of the encoder rows, sum_s w[s, n, t] dctx[s, n, c] (what mr_attn_denc computes
                           //             after the per-step loop), or null
  u64* xch;
  unsigned* status;
  int S, N, T, Ep, nbg;
  int xmap;                // block -> role map that puts a group's 32 slices on one XCD (decode_roles)
  unsigned hello_off;      // byte offset (from xch) of the XCC-id exchange
};

template <int R>
struct XchB {
  static constexpr int SPS = DG / R;
  static constexpr unsigned XD_SLOT = DG * R * DEPMAX * 8;      // partial dctx: [producer][sample][channel] {f32, tag}
  static constexpr unsigned XW_SLOT = R * SPS * DT * 8;         // partial dw:   [sample][share][position]
  static constexpr unsigned XA_SLOT = DG * R * DH * 8;          // partial dh_a: [producer][sample][unit]
  static constexpr unsigned XD_OFF = 0, XW_OFF = 2 * XD_SLOT, XA_OFF = XW_OFF + 2 * XW_SLOT;
  static constexpr unsigned GROUP = XA_OFF + 2 * XA_SLOT;
  static constexpr int CNT_D = R == 4 ? 5 : 11;                 // producers per thread of the dctx reduce (see PGA below)
};

constexpr int GLD = 64 + 8;   // LDS row stride (elements) of the two K = 64 MFMA operand tiles

template <int R>
struct LdsB {
  static constexpr int STG = 0;                                   // f32 [4 waves][9 tiles][R][16]: partial sums on their way out
  static constexpr int DGT = STG + 4 * 9 * R * 16 * 4;            // bf16 [16][GLD]  dgi  (r | z | n | 0)
  static constexpr int DHCT = DGT + 16 * GLD * 2;                 // bf16 [16][GLD]  dhproj | dgh r | z | n
  static constexpr int REDA = DHCT + 16 * GLD * 2;                // f32 [32/R][R][16]
  static constexpr int REDD = REDA + 512 * 4;                     // f32 [256][2]
  static constexpr int DCTX = REDD + 512 * 4;                     // f32 [288]
  static constexpr int DWP = DCTX + 288 * 4;                      // f32 [4][64]
  static constexpr int DE = DWP + 256 * 4;                        // f32 [R][64]
  static constexpr int HP = DE + R * 64 * 4;                      // f32 [R][16]
  static constexpr int DEAD = HP + R * 16 * 4;                    // int [4]
  static constexpr int ENC = DEAD + 16;                           // bf16 [T][4 * upp]
  static size_t bytes(int T, int Ep) {
    const int upp = cdiv(Ep / 4, DG / R);
    return (size_t)ENC + (size_t)T * upp * 8;
  }
};

}  // namespace

template <int R>
__global__ __launch_bounds__(256, 1) void decode_bwd_persist_kernel(DecB a) {
  typedef Mma<bf16_t>::Frag Frag;
  typedef XchB<R> X;
  typedef LdsB<R> L;
  constexpr int SPS = DG / R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sStg = (float*)(smem + L::STG) + wave_of_thread() * 9 * R * 16;   // this wave's staging tile
  bf16_t* sDG = (bf16_t*)(smem + L::DGT);
  bf16_t* sDHC = (bf16_t*)(smem + L::DHCT);
  float* sRedA = (float*)(smem + L::REDA);
  float* sRedD = (float*)(smem + L::REDD);
  float* sDctx = (float*)(smem + L::DCTX);
  float* sDwP = (float*)(smem + L::DWP);
  float* sDe = (float*)(smem + L::DE);
  float* sHp = (float*)(smem + L::HP);
  int* sDead = (int*)(smem + L::DEAD);
  bf16_t* sEnc = (bf16_t*)(smem + L::ENC);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  int g, bg;
  if (!decode_roles(a.xmap, a.nbg, g, bg)) return;
  const int T = a.T, Ep = a.Ep, N = a.N, S = a.S;
  const int nu = Ep >> 2;
  const int nloc = g / SPS, cpart = g % SPS, row_o = bg * R + nloc;
  const bool row_o_ok = row_o < N;
  const int upp = (nu + SPS - 1) / SPS;
  const int u0 = cpart * upp;
  const int nown = max(0, min(upp, nu - u0));

  // ---- weights: VGPR-resident MFMA fragments, rows = OUTPUT (context channel / hidden unit), K = this slice's 48 / 64 columns
  Frag wicT[9][2], wcatT[8][2];
  {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const int kc = g * DU + (lg & 1) * 8;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int ch = (wave + 4 * i) * 16 + l15;
      const bf16_t* p = a.ic_wt + (long long)ch * a.ldict + kc;
      uint4 t0 = z4, t1 = z4;
      if (ch < Ep) {
        t0 = *(const uint4*)(p + (lg < 2 ? 0 : DH));
        if (lg < 2) t1 = *(const uint4*)(p + 2 * DH);
      }
      wicT[i][0] = *(const Frag*)&t0;
      wicT[i][1] = *(const Frag*)&t1;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int u = (wave + 4 * i) * 16 + l15;
      const bf16_t* p = a.cat_wt + (long long)u * 4 * DH + kc;
      wcatT[i][0] = *(const Frag*)(p + (lg < 2 ? 0 : DH));
      wcatT[i][1] = *(const Frag*)(p + (lg < 2 ? 2 * DH : 3 * DH));
    }
  }
  // tanh chain ownership: wave <-> sample, lane <-> (unit tq of the slice, chunk tc of 16 positions).  The thread's 16 eproj
  // elements and its 16 deproj accumulators live in registers for the whole sequence.
  static_assert(R == 4, "the backward kernel is built for groups of 4 rows");
  const int tq = lane & 15, tc = lane >> 4;
  const float vq = a.v[g * DU + tq];
  float evr[16], dacc[16];
  {
    const int r = bg * R + wave;
    unsigned short raw[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)      // (all sixteen in flight: selected against 0 inside this loop they were 16 round trips in a row)
      raw[j] = ((const unsigned short*)a.eproj)[((long long)min(r, N - 1) * T + min(tc * 16 + j, T - 1)) * DH + g * DU + tq];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      evr[j] = (r < N && tc * 16 + j < T) ? bf16_lo(raw[j]) : 0.f;
      dacc[j] = 0.f;
    }
  }
  float dvq = 0.f;
  // denc of this slice's (sample, channel share): thread <-> (position lane, channel group wave of 18), over all steps
  float dnc[18];
#pragma unroll
  for (int j = 0; j < 18; ++j) dnc[j] = 0.f;
  // ---- LDS residents
  for (int i = tid; i < T * upp; i += 256) {
    const int t = i / upp, u = i - t * upp;
    uint2 e = make_uint2(0, 0);
    if (row_o_ok && u < nown) e = *(const uint2*)(a.enc + ((long long)row_o * T + t) * Ep + (u0 + u) * 4);
    *(uint2*)(sEnc + (long long)i * 4) = e;
  }
  for (int i = tid; i < 2 * 16 * GLD / 8; i += 256) ((uint4*)sDG)[i] = make_uint4(0, 0, 0, 0);   // sDG and sDHC (adjacent)
  for (int i = tid; i < 288; i += 256) sDctx[i] = 0.f;
  if (tid < 4) sDead[tid] = 0;

  const rsrc_t rx = make_rsrc(a.xch);
  const unsigned xg = (unsigned)bg * X::GROUP;
  const bool local = group_on_one_xcd(rx, a.hello_off + (unsigned)(bg * DG * 16), g, a.xmap, &sDead[3], a.status);
  // GRU ownership: thread <-> (sample gm, unit gu of the slice)
  const int gm = tid >> 4, gu = tid & 15, row_g = bg * R + gm, jg = g * DU + gu;
  const bool gru_thread = gm < R;
  const bool row_g_ok = gru_thread && row_g < N;
  // edge C consumer: (producer group pg, sample cm, unit pair p8) -> R producers each
  const int p8 = tid & 7, cm = (tid >> 3) & (R - 1), pgc = tid / (8 * R);
  unsigned offA[R];
#pragma unroll
  for (int k = 0; k < R; ++k) offA[k] = (unsigned)((((pgc + SPS * k) * R + cm) * DH + g * DU + p8 * 2) * 8);
  // edge A consumer: (producer group pga, channel pair pr of the share) -> up to CNT_D producers
  const int npair = 2 * upp, PGA = 256 / npair;
  const int pr = tid % npair, pga = tid / npair;
  unsigned offD[X::CNT_D], wantD = 0;
#pragma unroll
  for (int k = 0; k < X::CNT_D; ++k) {
    const int prod = pga + PGA * k;
    const bool wanted = pga < PGA && prod < DG && pr < 2 * nown;
    offD[k] = wanted ? (unsigned)(((prod * R + nloc) * DEPMAX + u0 * 4 + pr * 2) * 8) : 0u;   // (every offset is loaded from)
    if (wanted) wantD |= 1u << k;
  }
  const int npga = min(PGA, DG);
  // edge B consumer: lane = (half, position pair tp); samples wave + 4i
  const int tp = lane & 31, half = lane >> 5;
  float dh_b = 0.f;
  bool dead = false;
  __syncthreads();
  // phase clock, as in the forward kernel
  const bool timing = tid == 0 && bg == 0 && g < 2 && a.status[2] == TIMING_MAGIC;
  unsigned long long tprev = timing ? wall_clock64() : 0ull;
  unsigned tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DEC_TICK(i)                                   \
  if (timing) {                                       \
    const unsigned long long tn_ = wall_clock64();    \
    tacc[i] += (unsigned)(tn_ - tprev);               \
    tprev = tn_;                                      \
  }

  // The operands of a step that depend on nothing of the chain (saved gates, h, the output layer's gradient, the softmax weights:
  // HBM latency) are fetched ONE ITERATION AHEAD, behind the GRU backward: in front of the first sweep of their own step they
  // delayed it (loads return in order).  UNCONDITIONAL loads, rows / positions clamped into the buffers, masked where they are
  // USED: a load under a branch (or selected against a constant right behind it) is waited for at the join.
  float n_sr = 0.f, n_sz = 0.f, n_sn = 0.f, n_wden = 0.f;
  unsigned short n_ghn = 0, n_hpv = 0, n_dhc = 0, n_hpj = 0;
  float2 n_wat[R / 4];
  const int tl = min(lane, T - 1), t0c = min(2 * tp, T - 1), t1c = min(2 * tp + 1, T - 1);
  auto fetch_ops = [&](int s1) {
    if (gru_thread) {          // (whole waves)
      const long long rn = (long long)s1 * N + min(row_g, N - 1);
      const float* sv = a.SAVE_all + rn * 3 * DH + jg;
      n_sr = sv[0]; n_sz = sv[DH]; n_sn = sv[2 * DH];
      const unsigned short* hc = (const unsigned short*)(a.HC_all + rn * 4 * DH);
      n_hpj = hc[jg];
      n_ghn = hc[DH + 2 * DH + jg];
      n_hpv = ((const unsigned short*)a.H_all)[rn * DH + jg];
      n_dhc = ((const unsigned short*)a.DHO_all)[rn * DH + jg];
    }
    n_wden = a.W_att[((long long)s1 * N + min(row_o, N - 1)) * T + tl];
#pragma unroll
    for (int i = 0; i < R / 4; ++i) {
      const float* wp = a.W_att + ((long long)s1 * N + min(bg * R + wave + 4 * i, N - 1)) * T;
      n_wat[i].x = wp[t0c];
      n_wat[i].y = wp[t1c];
    }
  };
  fetch_ops(S - 1);
  for (int it = 0; it < S; ++it) {
    const int s = S - 1 - it;
    const unsigned tag = (unsigned)(it + 1);
    const unsigned slot = (unsigned)(it & 1);
    // ---- operands that depend on nothing of the chain (fetched during the PREVIOUS iteration, see fetch_ops)
    const float sr = n_sr, sz = n_sz, sn = n_sn;
    const unsigned short ghn = n_ghn, hpv = n_hpv, dhc = n_dhc, hpj = n_hpj;
    const float wden = n_wden;
    const bool wden_ok = a.denc && row_o_ok && lane < T;
    float2 wat[R / 4];
#pragma unroll
    for (int i = 0; i < R / 4; ++i) wat[i] = n_wat[i];
    // ---- edge C of the previous iteration: dh_a of the own units = sum over the 32 producers
    float dh_a = 0.f;
    if (it > 0) {
      u32x4 av[R];
      if (!dead && !gather_pairs<R>(rx, offA, xg + X::XA_OFF + (slot ^ 1u) * X::XA_SLOT, (1u << R) - 1u, tag - 1u, av)) {
        dead = true;
        if (lane == 0) { atomicMax(a.status, 4u); sDead[0] = 1; }
      }
      float s0 = 0.f, s1 = 0.f;
      if (!dead) {
#pragma unroll
        for (int k = 0; k < R; ++k) { s0 += __uint_as_float(av[k][0]); s1 += __uint_as_float(av[k][2]); }
      }
      *(float2*)(sRedA + (pgc * R + cm) * DU + p8 * 2) = make_float2(s0, s1);
      lds_barrier();
      if (gru_thread) {
#pragma unroll
        for (int k = 0; k < SPS; ++k) dh_a += sRedA[(k * R + gm) * DU + gu];
      }
    }
    DEC_TICK(0)
    // ---- GRU backward of the own units
    if (gru_thread) {
      const float hn = bf16_lo(ghn), hp = bf16_lo(hpv);
      float gsum = dh_a + dh_b + bf16_lo(dhc);
      if (sDead[0] | sDead[1] | sDead[2]) gsum = __builtin_nanf("");
      const float dn = gsum * (1.f - sz);
      const float dz = gsum * (hp - sn);
      const float dpre_n = dn * (1.f - sn * sn);
      const float dr = dpre_n * hn;
      const float dpre_r = dr * sr * (1.f - sr);
      const float dpre_z = dz * sz * (1.f - sz);
      dh_b = gsum * sz;
      const float vm = row_g_ok ? 1.f : 0.f;          // rows beyond the batch carry the clamped row's operands: keep them out
      const bf16_t b_r = (bf16_t)(vm * dpre_r), b_z = (bf16_t)(vm * dpre_z), b_n = (bf16_t)(vm * dpre_n),
                   b_nr = (bf16_t)(vm * dpre_n * sr);
      bf16_t* dg = sDG + gm * GLD;
      dg[gu] = b_r; dg[16 + gu] = b_z; dg[32 + gu] = b_n;
      bf16_t* dc = sDHC + gm * GLD;
      dc[16 + gu] = b_r; dc[32 + gu] = b_z; dc[48 + gu] = b_nr;
      sHp[gm * DU + gu] = bf16_lo(hpj);
      if (row_g_ok) {
        const long long rn = (long long)s * N + row_g;
        bf16_t* o1 = a.DGI_all + rn * 3 * DH + jg;
        o1[0] = b_r; o1[DH] = b_z; o1[2 * DH] = b_n;
        bf16_t* o2 = a.DHC_all + rn * 4 * DH + DH + jg;
        o2[0] = b_r; o2[DH] = b_z; o2[2 * DH] = b_nr;
      }
    }
    if (s > 0) fetch_ops(s - 1);       // a whole iteration to arrive
    lds_barrier();
    DEC_TICK(1)
    // ---- partial dctx of every sample over the own 48 gate units -> edge A
    {
      const Frag a0 = *(const Frag*)(sDG + l15 * GLD + lg * 8);
      const Frag a1 = *(const Frag*)(sDG + l15 * GLD + 32 + lg * 8);
      const unsigned xd = xg + X::XD_OFF + slot * X::XD_SLOT + (unsigned)(((g * R + l15) * DEPMAX + lg * 4) * 8);
      // only the R rows of a 16-row MFMA tile are samples: the tiles go through the wave's staging area so that ALL lanes store
      // (9 tiles x R rows x 8 pairs = 288 pairs = 5 store instructions a wave instead of 18 by a quarter of the lanes)
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        Mma<bf16_t>::run(acc, wicT[i][0], a0);
        Mma<bf16_t>::run(acc, wicT[i][1], a1);
        if (l15 < R) *(f32x4*)(sStg + (i * R + l15) * 16 + lg * 4) = acc;
      }
      (void)xd;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes, then its reads (no other wave involved)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int pp = lane + 64 * k;                    // pair index: [tile][row][pair of 8]
        const int ti = pp >> 5, rw = (pp >> 3) & 3, pr8 = pp & 7;
        const int c0 = (wave + 4 * ti) * 16 + pr8 * 2;
        if (pp < 288 && c0 < Ep) {
          const float2 v2 = *(const float2*)(sStg + pp * 2);
          gran2_publish(rx, xg + X::XD_OFF + slot * X::XD_SLOT + (unsigned)(((g * R + rw) * DEPMAX + c0) * 8),
                        __float_as_uint(v2.x), __float_as_uint(v2.y), tag, local);
        }
      }
    }
    DEC_TICK(2)
    // ---- reduce this slice's channel share of dctx[sample nloc] over the producers
    {
      u32x4 dvv[X::CNT_D];
      if (!dead && !gather_pairs<X::CNT_D>(rx, offD, xg + X::XD_OFF + slot * X::XD_SLOT, wantD, tag, dvv)) {
        dead = true;
        if (lane == 0) { atomicMax(a.status, 5u); sDead[1] = 1; }
      }
      DEC_TICK(3)
      float s0 = 0.f, s1 = 0.f;
      if (!dead) {
#pragma unroll
        for (int k = 0; k < X::CNT_D; ++k)
          if ((wantD >> k) & 1u) { s0 += __uint_as_float(dvv[k][0]); s1 += __uint_as_float(dvv[k][2]); }
      }
      *(float2*)(sRedD + tid * 2) = make_float2(s0, s1);
      lds_barrier();
      if (tid < 4 * nown) {
        float c = 0.f;
#pragma unroll 8
        for (int k = 0; k < npga; ++k) c += sRedD[k * 2 * npair + tid];
        const bf16_t cb = (bf16_t)c;
        sDctx[tid] = (float)cb;
        if (row_o_ok) a.DCTX_all[((long long)s * N + row_o) * Ep + u0 * 4 + tid] = cb;
      }
      lds_barrier();
    }
    DEC_TICK(4)
    // ---- partial dw[t] = dctx[share] . enc[t, share]  (wave = unit quarter, lane = position) -> edge B
    if (a.denc) {
      const float wd = wden_ok ? wden : 0.f;
#pragma unroll
      for (int j = 0; j < 18; j += 2) {
        const float2 dc = *(const float2*)(sDctx + wave * 18 + j);
        dnc[j] += wd * dc.x;
        dnc[j + 1] += wd * dc.y;
      }
    }
    {
      const int uq = (nown + 3) >> 2, ub = wave * uq, ue = min(nown, ub + uq);
      float d = 0.f;
      if (lane < T) {
        const bf16_t* er = sEnc + lane * upp * 4;
        for (int u = ub; u < ue; ++u) {
          const uint2 ev = *(const uint2*)(er + u * 4);
          const f32x4 dc = *(const f32x4*)(sDctx + u * 4);
          d += dc[0] * bf16_lo(ev.x) + dc[1] * bf16_hi(ev.x) + dc[2] * bf16_lo(ev.y) + dc[3] * bf16_hi(ev.y);
        }
      }
      sDwP[wave * 64 + lane] = d;
      lds_barrier();
      if (tid < 32) {
        float d0 = (sDwP[2 * tid] + sDwP[64 + 2 * tid]) + (sDwP[128 + 2 * tid] + sDwP[192 + 2 * tid]);
        float d1 = (sDwP[2 * tid + 1] + sDwP[64 + 2 * tid + 1]) + (sDwP[128 + 2 * tid + 1] + sDwP[192 + 2 * tid + 1]);
        if (a.ga && cpart == 0 && row_o_ok) {
          const float* gp = a.ga + (long long)row_o * a.ldga + (long long)s * T;
          if (2 * tid < T) d0 += gp[2 * tid];
          if (2 * tid + 1 < T) d1 += gp[2 * tid + 1];
        }
        gran2_publish(rx, xg + X::XW_OFF + slot * X::XW_SLOT + (unsigned)(((nloc * SPS + cpart) * DT + 2 * tid) * 8),
                    __float_as_uint(d0), __float_as_uint(d1), tag, local);
      }
    }
    DEC_TICK(5)
    // ---- dw of every sample = sum over its channel shares; softmax backward
    {
      constexpr int NB = (R / 4) * (SPS / 2);
      unsigned offW[NB];
      u32x4 wv[NB];
#pragma unroll
      for (int i = 0; i < R / 4; ++i)
#pragma unroll
        for (int k = 0; k < SPS / 2; ++k)
          offW[i * (SPS / 2) + k] = (unsigned)((((wave + 4 * i) * SPS + half * (SPS / 2) + k) * DT + 2 * tp) * 8);
      if (!dead && !gather_pairs<NB>(rx, offW, xg + X::XW_OFF + slot * X::XW_SLOT, (1u << NB) - 1u, tag, wv)) {
        dead = true;
        if (lane == 0) { atomicMax(a.status, 6u); sDead[2] = 1; }
      }
      DEC_TICK(6)
#pragma unroll
      for (int i = 0; i < R / 4; ++i) {
        float d0 = 0.f, d1 = 0.f;
        if (!dead) {
#pragma unroll
          for (int k = 0; k < SPS / 2; ++k) {
            d0 += __uint_as_float(wv[i * (SPS / 2) + k][0]);
            d1 += __uint_as_float(wv[i * (SPS / 2) + k][2]);
          }
        }
        d0 += __shfl_xor(d0, 32, 64);
        d1 += __shfl_xor(d1, 32, 64);
        // both halves of the wave hold the same pairs: the wave sum counts every position twice
        const bool rv = bg * R + wave + 4 * i < N;
        const float w0 = (rv && 2 * tp < T) ? wat[i].x : 0.f, w1 = (rv && 2 * tp + 1 < T) ? wat[i].y : 0.f;
        const float dot = 0.5f * wave_sum_dpp(w0 * d0 + w1 * d1);
        if (half == 0) *(float2*)(sDe + (wave + 4 * i) * 64 + 2 * tp) = make_float2(w0 * (d0 - dot), w1 * (d1 - dot));
      }
      lds_barrier();
    }
    DEC_TICK(7)
    // ---- tanh chain of the own units: dhproj (reduced over the positions), deproj / dv (accumulated over the steps)
    {
      const int n = wave;
      const float hp = sHp[n * DU + tq];
      float dhp = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < 16; j4 += 4) {
        const f32x4 de4 = *(const f32x4*)(sDe + n * 64 + tc * 16 + j4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float th = dec_tanh(hp + evr[j4 + e]);
          const float gg = de4[e] * vq * (1.f - th * th);
          dacc[j4 + e] += gg;
          dvq += de4[e] * th;
          dhp += gg;
        }
      }
      dhp += __shfl_xor(dhp, 16, 64);
      dhp += __shfl_xor(dhp, 32, 64);
      if (lane < DU) {
        const bf16_t hb = (bf16_t)dhp;
        sDHC[n * GLD + lane] = hb;
        const int r = bg * R + n;
        if (r < N) a.DHC_all[((long long)s * N + r) * 4 * DH + g * DU + lane] = hb;
      }
    }
    DEC_TICK(8)
    if (s == 0) break;
    lds_barrier();
    // ---- partial dh_a of every sample over the own 64 stacked columns -> edge C
    {
      const Frag a0 = *(const Frag*)(sDHC + l15 * GLD + lg * 8);
      const Frag a1 = *(const Frag*)(sDHC + l15 * GLD + 32 + lg * 8);
      const unsigned xa = xg + X::XA_OFF + slot * X::XA_SLOT + (unsigned)(((g * R + l15) * DH + lg * 4) * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        Mma<bf16_t>::run(acc, wcatT[i][0], a0);
        Mma<bf16_t>::run(acc, wcatT[i][1], a1);
        if (l15 < R) *(f32x4*)(sStg + (i * R + l15) * 16 + lg * 4) = acc;
      }
      (void)xa;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pp = lane + 64 * k;                    // 8 tiles x R rows x 8 pairs = 256 pairs
        const int ti = pp >> 5, rw = (pp >> 3) & 3, pr8 = pp & 7;
        const float2 v2 = *(const float2*)(sStg + pp * 2);
        gran2_publish(rx, xg + X::XA_OFF + slot * X::XA_SLOT + (unsigned)(((g * R + rw) * DH + (wave + 4 * ti) * 16 + pr8 * 2) * 8),
                      __float_as_uint(v2.x), __float_as_uint(v2.y), tag, local);
      }
    }
    DEC_TICK(9)
  }
  if (timing) {
#pragma unroll
    for (int i = 0; i < 10; ++i) a.status[8 + 16 * g + i] = tacc[i];
  }
#undef DEC_TICK
  // ---- the accumulated gradients of eproj (own 16 columns) and v
  {
    const int r = bg * R + wave;
    if (r < N) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int t = tc * 16 + j;
        if (t < T) a.deproj[((long long)r * T + t) * DH + g * DU + tq] = dacc[j];
      }
    }
    if (a.denc && row_o_ok && lane < T) {
      bf16_t* dp = a.denc + ((long long)row_o * T + lane) * Ep + u0 * 4 + wave * 18;
#pragma unroll
      for (int j = 0; j < 18; j += 2)
        if (wave * 18 + j < 4 * nown) *(unsigned*)(dp + j) = pack_bf16(dnc[j], dnc[j + 1]);
    }
    dvq += __shfl_xor(dvq, 16, 64);
    dvq += __shfl_xor(dvq, 32, 64);
    if (lane < DU) atomicAdd(a.dv + g * DU + lane, dvq);
  }
}

namespace {
// rows per batch group: 4 while the grid (32 workgroups per group, ONE per CU: 460+ registers a lane) stays within the 256 CUs,
// else 8.  The bytes a workgroup gathers per hand-off scale with the rows: before the XCD placement 4-row groups ran a step in
// 8.6 us forward / 13.5 us backward, 8-row groups in 10.6 / 21 us (the backward then spills; it is built for 4 rows only).  All
// workgroups of a group must be co-resident; workgroups that are dispatched late (a kernel of another stream still holds their CU)
// only delay their group -- every wait is bounded (tests/test_decode_persist_gpu.py: beside a busy side stream).
int decode_rows(int N) { return N <= 32 ? 4 : 8; }
// CUs of the current device (cached per device): every workgroup of a launch must be resident at once, one per CU
int decode_cus() {
  static int cus[MR_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MR_MAX_DEVICES) return 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus[dev] = n > 0 ? n : -1;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}
// the XCD-colocating block map assumes the whole chip: 8 XCDs x 32 CUs.  decode_persist = 2: never use it (A/B knob)
int decode_xmap(int nbg) { return (MR_TUNE(decode_persist) != 2 && nbg <= 8 && decode_cus() == 8 * DG) ? 1 : 0; }
unsigned decode_group_bytes(int R) { return R == 4 ? Xch<4>::GROUP : Xch<8>::GROUP; }
long long decode_ws_bytes(int N) {
  const int R = decode_rows(N);
  return (long long)cdiv(N, R) * decode_group_bytes(R) + HELLO_BYTES + 256;
}

template <int R>
int decode_launch(const DecP& a, hipStream_t stream) {
  static bool attr_set[MR_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= 0 && dev < MR_MAX_DEVICES && !attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)decode_fwd_persist_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)Lds<R>::bytes(DT, DEPMAX)) != hipSuccess) {
      set_error("mr_decode_persist_fwd: cannot raise the dynamic LDS limit");
      return MR_ERR_LAUNCH;
    }
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(decode_fwd_persist_kernel<R>, dim3((a.xmap ? 8 : a.nbg) * DG), dim3(256), Lds<R>::bytes(a.T, a.Ep), stream, a);
  MR_CHECK_LAUNCH();
  return MR_OK;
}
unsigned decode_bwd_group_bytes(int) { return XchB<4>::GROUP; }
long long decode_bwd_ws_bytes(int N) {
  const int R = decode_rows(N);
  return (long long)cdiv(N, R) * decode_bwd_group_bytes(R) + HELLO_BYTES + 256;
}

template <int R>
int decode_bwd_launch(const DecB& a, hipStream_t stream) {
  static bool attr_set[MR_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= 0 && dev < MR_MAX_DEVICES && !attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)decode_bwd_persist_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LdsB<R>::bytes(DT, DEPMAX)) != hipSuccess) {
      set_error("mr_decode_persist_bwd: cannot raise the dynamic LDS limit");
      return MR_ERR_LAUNCH;
    }
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(decode_bwd_persist_kernel<R>, dim3((a.xmap ? 8 : a.nbg) * DG), dim3(256), LdsB<R>::bytes(a.T, a.Ep), stream, a);
  MR_CHECK_LAUNCH();
  return MR_OK;
}
}  // namespace

extern "C" {

// host only: can the persistent decode forward take this shape on the current device?  (bf16, H = 512, N <= 64, T <= 64,
// Ep <= 576 a multiple of 8, ceil(N / rows) * 32 <= CUs)
int mr_decode_persist_ok(int dtype, int N, int T, int H, int Ep) {
  if (!(dtype == MR_BF16 && H == DH && N >= 1 && N <= 64 && T >= 1 && T <= DT && Ep >= 8 && Ep <= DEPMAX && Ep % 8 == 0 &&
        MR_TUNE(decode_persist) != 0))
    return 0;
  // one workgroup per CU, all of them resident at once (a partitioned or smaller device falls back to the per-step launches)
  return cdiv(N, decode_rows(N)) * DG <= decode_cus() ? 1 : 0;
}

long long mr_decode_persist_ws_bytes(int N) { return decode_ws_bytes(N); }

// All S steps of the attention-GRU decode loop in one launch (see the file header).  idx [S][N] = the word fed to each step;
// H_all[0] = the initial state.  flags null: every step is fed idx[s].  flags [S] (device): where flags[s] == 0, step s + 1 is fed
// the arg-max of step s's output layer out_w [C][H] / out_b (C <= 256) instead, and the kernel WRITES that word to idx[s + 1].
// ws_bytes negative: the caller already zeroed the workspace.
int mr_decode_persist_fwd(const void* cat_w, const float* cat_b, const void* ic_w, long long ldic, const void* G, long long ldG,
                          long long* idx, const int* flags, const void* out_w, const float* out_b, int C, const void* eproj,
                          const void* enc, const float* v, void* H_all, void* HC_all, float* W_att, void* CTX_all,
                          float* SAVE_all, void* ws, long long ws_bytes, int S, int N, int T, int Ep, hipStream_t stream) {
  const bool prezeroed = ws_bytes < 0;
  if (prezeroed) ws_bytes = -ws_bytes;
  MR_CHECK_ARG(S >= 1 && N >= 1 && N <= 64 && T >= 1 && T <= DT && Ep >= 8 && Ep <= DEPMAX && Ep % 8 == 0 && ldic >= Ep &&
                   ldG >= 3 * DH,
               "mr_decode_persist_fwd: bad shape S=%d N=%d T=%d Ep=%d", S, N, T, Ep);
  MR_CHECK_ARG(flags == nullptr || (out_w != nullptr && C >= 1 && C <= 8 * DG),
               "mr_decode_persist_fwd: arg-max feedback needs the output layer and C <= 256 (C=%d)", C);
  MR_CHECK_ARG(ws_bytes >= decode_ws_bytes(N), "mr_decode_persist_fwd: workspace too small (%lld < %lld)", ws_bytes,
               decode_ws_bytes(N));
  const int R = decode_rows(N), nbg = cdiv(N, R);
  if (!prezeroed && hipMemsetAsync(ws, 0, (size_t)decode_ws_bytes(N), stream) != hipSuccess) {
    set_error("mr_decode_persist_fwd: memset of the exchange buffer failed");
    return MR_ERR_LAUNCH;
  }
  DecP a{(const bf16_t*)cat_w, cat_b, (const bf16_t*)ic_w, ldic, (const bf16_t*)G, ldG, idx, flags, (const bf16_t*)out_w, out_b, C,
         (const bf16_t*)eproj,
         (const bf16_t*)enc, v, (bf16_t*)H_all, (bf16_t*)HC_all, W_att, (bf16_t*)CTX_all, SAVE_all, (u64*)ws,
         (unsigned*)((char*)ws + (long long)nbg * decode_group_bytes(R) + HELLO_BYTES), S, N, T, Ep, nbg, decode_xmap(nbg),
         (unsigned)(nbg * decode_group_bytes(R))};
  return R == 4 ? decode_launch<4>(a, stream) : decode_launch<8>(a, stream);
}

// host only: the persistent BACKWARD of the loop (groups of 4 rows only: N <= 32)
int mr_decode_persist_bwd_ok(int dtype, int N, int T, int H, int Ep) {
  return (N <= 32 && mr_decode_persist_ok(dtype, N, T, H, Ep)) ? 1 : 0;
}

long long mr_decode_persist_bwd_ws_bytes(int N) { return N <= 32 ? decode_bwd_ws_bytes(N) : 0; }

// All S steps of the decode loop's backward (in reverse) in one launch -- what the per-step mr_gemm_gru_bwd / mr_gru_bwd2 +
// mr_gemm_nt + mr_attn_bwd2 launches compute.  cat_wt [H][4H] and ic_wt [Ep][ldict >= 3H] are the TRANSPOSED weight images
// (row = output of the backward GEMM); DHO_all [S][N][H] = gradient of every h' from the output layer; ga (nullable) = gradient
// of the attention weights, element (n, s, t) at ga[n * ldga + s * T + t].  Writes DGI_all [S][N][3H], DHC_all [S][N][4H],
// DCTX_all [S][N][Ep], deproj [N][T][H] (f32, plain stores), ADDS into dv [H] (f32) and, when denc is not null, writes
// denc [N][T][Ep] = sum_s W_att[s] dctx[s] (mr_attn_denc's result).  ws as mr_decode_persist_fwd (mr_decode_persist_bwd_ws_bytes).
int mr_decode_persist_bwd(const void* cat_wt, const void* ic_wt, long long ldict, const void* eproj, const void* enc,
                          const float* v, const void* H_all, const void* HC_all, const float* W_att, const float* SAVE_all,
                          const void* DHO_all, const float* ga, long long ldga, void* DGI_all, void* DHC_all, void* DCTX_all,
                          float* deproj, float* dv, void* denc, void* ws, long long ws_bytes, int S, int N, int T, int Ep,
                          hipStream_t stream) {
  const bool prezeroed = ws_bytes < 0;
  if (prezeroed) ws_bytes = -ws_bytes;
  MR_CHECK_ARG(S >= 1 && N >= 1 && N <= 32 && T >= 1 && T <= DT && Ep >= 8 && Ep <= DEPMAX && Ep % 8 == 0 && ldict >= 3 * DH,
               "mr_decode_persist_bwd: bad shape S=%d N=%d T=%d Ep=%d", S, N, T, Ep);
  MR_CHECK_ARG(ws_bytes >= decode_bwd_ws_bytes(N), "mr_decode_persist_bwd: workspace too small (%lld < %lld)", ws_bytes,
               decode_bwd_ws_bytes(N));
  const int R = decode_rows(N), nbg = cdiv(N, R);
  if (!prezeroed && hipMemsetAsync(ws, 0, (size_t)decode_bwd_ws_bytes(N), stream) != hipSuccess) {
    set_error("mr_decode_persist_bwd: memset of the exchange buffer failed");
    return MR_ERR_LAUNCH;
  }
  DecB a{(const bf16_t*)cat_wt, (const bf16_t*)ic_wt, ldict, (const bf16_t*)eproj, (const bf16_t*)enc, v,
         (const bf16_t*)H_all, (const bf16_t*)HC_all, W_att, SAVE_all, (const bf16_t*)DHO_all, ga, ldga, (bf16_t*)DGI_all,
         (bf16_t*)DHC_all, (bf16_t*)DCTX_all, deproj, dv, (bf16_t*)denc, (u64*)ws,
         (unsigned*)((char*)ws + (long long)nbg * decode_bwd_group_bytes(R) + HELLO_BYTES), S, N, T, Ep, nbg, decode_xmap(nbg),
         (unsigned)(nbg * decode_bwd_group_bytes(R))};
  return decode_bwd_launch<4>(a, stream);
}

}  // extern "C"

}  // namespace mr
