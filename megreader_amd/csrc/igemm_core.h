// MFMA implicit-GEMM cores for gfx950.
//
//  igemm_nt_kernel : C[M,N]  = Agather[M,K] * B[N,K]^T   (conv fwd, conv dgrad, linear, LSTM x-proj / steps)
//  igemm_tn_kernel : C[NA,NB] += sum_p A[p,NA] * Bgather[p,NB]   (conv wgrad, linear/LSTM weight grads)
//
// Layout contract: activations are NHWC (channel-contiguous), weights are [N][K] with K contiguous
// (KRSC for convolutions).  One 16-byte vector (8 bf16 / 4 f32) is the unit of every global and LDS
// access.  Workgroup = 256 threads = 4 waves (2x2), wave = 64 lanes.
//
// NT LDS image: [8 k-chunks][ROWS][16 B], slot(row, kc) = kc*ROWS + (row ^ kc).  The XOR keeps
//   * ds_write_b128 conflict-free (an 8-lane group writes one row x 8 chunks -> 8 distinct 16-B bank groups)
//   * ds_read_b128 conflict-free (each 16-lane service group of the MFMA fragment read touches
//     16 rows that stay distinct mod 16).
// The MFMA is issued with swapped operands (weights as the A operand) so each lane ends up with
// 4 consecutive output channels of one pixel -> 8-byte (bf16) / 16-byte (f32) row-major stores.
//
// TN LDS image: [BP rows(p)][128 cols], 32-byte column pieces XOR-swizzled by a hash of the row so
// that ds_read_b64_tr_b16 (the gfx950 LDS transpose read that turns "8 consecutive p at one column"
// into an MFMA K-fragment) is conflict-free across a 32-lane half.
#pragma once
#include "common.h"
#include <type_traits>

namespace mr {

struct ConvGeom {
  int Hg, Wg, Cg, ldg;  // gathered tensor: spatial dims, channels per tap, pixel stride (elements)
  int Hm, Wm;           // spatial dims of the row (M / P) index space
  int R, S, sh, sw, ph, pw, dh, dw;
  int mode;             // 1: forward gather   hi = hm*sh - ph + r*dh
                        // 2: dgrad gather     hi = (hm + ph - r*dh) / sh   (must divide exactly)
};

__device__ __forceinline__ bool conv_src(const ConvGeom& g, int hm, int wm, int r, int s, int& hi, int& wi) {
  if (g.mode == 1) {
    hi = hm * g.sh - g.ph + r * g.dh;
    wi = wm * g.sw - g.pw + s * g.dw;
  } else {
    int hn = hm + g.ph - r * g.dh;
    int wn = wm + g.pw - s * g.dw;
    if (hn < 0 || wn < 0) return false;
    if (g.sh == 1 && g.sw == 1) {   // uniform fast path: every stride-1 layer skips two runtime integer divisions per
      hi = hn;                      // (row, tap) of the tile prologue (18 per staged row for a 3x3 filter)
      wi = wn;
    } else {
      hi = hn / g.sh;
      wi = wn / g.sw;
      if (hi * g.sh != hn || wi * g.sw != wn) return false;
    }
  }
  return (unsigned)hi < (unsigned)g.Hg && (unsigned)wi < (unsigned)g.Wg;
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 Frag;
  static __device__ __forceinline__ void run(f32x4& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // exact-f32 MFMA (v_mfma_f32_16x16x4_f32).  A 16-byte fragment holds k = 4*(lane>>4)+j, j=0..3;
  // A and B use the same k permutation so the sum is unchanged.
  typedef f32x4 Frag;
  static __device__ __forceinline__ void run(f32x4& acc, const Frag& a, const Frag& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
  }
};

struct NtArgs {
  const void* A;
  const void* B;
  int M, N, K;
  long long lda;  // dense A row stride (elements); ignored in conv mode
  int ldb;        // B row stride (elements)
  const void* zero;  // >= 16 bytes of zeros in global memory (source of padded / out-of-range vectors)
  int m_begin = 0;   // first row this launch computes (rows [m_begin, M)): lets the host cut a problem into a head that
                     // fills the CUs in whole rounds of big tiles and a small tail (glds / big kernels only)
  // Split reduction (igemm_nt_glds_kernel, bf16): gridDim.y = ksplit workgroups share one output tile, each runs 1/ksplit of
  // the k-steps, writes its f32 partial tile to a slab of `ws` and takes a ticket; the last arriver adds the slabs IN SPLIT
  // ORDER (same bits whoever is last) and runs the epilogue.  For launches of a few tiles with a long reduction (the 512-row
  // layers of a batch-32 / batch-2 step: 64 tiles x 72 dependent k-steps on a quarter of the CUs).
  // ws = [NT_SPLIT_TICKETS ints, zero between launches][slabs of BM*BN*4 bytes, index tile*ksplit + split].
  int ksplit = 1;
  void* ws = nullptr;
  // Rows of the M index space one tile COVERS (igemm_nt_big_kernel only; 0 = the tile height BM): a tile of BM rows computes
  // rows [m_begin + t * tile_rows, + tile_rows) and leaves its last BM - tile_rows rows empty.  Lets tiles start on image
  // boundaries whatever the tile height (the pooled epilogue, EpiPool: 2 images of 4 x 33 = 264 rows in a 272-row tile).
  int tile_rows = 0;
};
constexpr int NT_SPLIT_TICKETS = 4096;

__device__ __forceinline__ uint4 ldg16(const void* p) { return *(const uint4*)p; }

// Column statistics of a wave's output tile, for epilogues that carry a `stats` pointer (EpiStore): the lane that owns
// the run of 4*CNT channels n.. of rows m = base + 16 j + l15 accumulates sum / sum of squares over its TM rows, the 16
// lanes of a row group (same lg) are combined with four butterfly steps, lane l15 == 0 adds the result to the f64
// accumulators.  2 * 4*CNT f64 atomics per (wave, lg): 512 per 128x128 tile against 16384 elements.
template <typename Epi> struct EpiHasStats { static constexpr bool value = false; };  // specialised for EpiStore below
template <typename Epi> struct EpiIsPool { static constexpr bool value = false; };   // specialised for EpiPool below
__device__ __forceinline__ float row16_sum(float v) {   // total over the lane's 16-lane DPP row, in every lane of it
#define MR_ROW_ROR(V, N) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x120 + (N), 0xf, 0xf, false))
  v += MR_ROW_ROR(v, 8);
  v += MR_ROW_ROR(v, 4);
  v += MR_ROW_ROR(v, 2);
  v += MR_ROW_ROR(v, 1);
#undef MR_ROW_ROR
  return v;
}
template <typename T, int CNT> struct EpiColStats {
  float s[4 * CNT], q[4 * CNT];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int e = 0; e < 4 * CNT; ++e) s[e] = q[e] = 0.f;
  }
  template <typename Epi> __device__ __forceinline__ void add(const Epi& epi, int m, int n, const f32x4* v) {
    if (m >= epi.M) return;
    if constexpr (Epi::kBnb) {
      // BACKWARD mode (EpiStoreB, mr_conv2d_dgrad_bnb): the stored matrix is the gradient dy of a training-mode BatchNorm's output.
      // s += g', q += g' * x with g' = the value the store writes (addend included), zeroed where the BatchNorm's fused ReLU
      // was off (y <= 0); flush() turns the raw sum into  sum g' * xhat = rstd * (q - mean * s)  -- mean / rstd stay out of
      // the loop, the cancellation happens once per partial sum in f64.
      const long long row = (long long)m * epi.ldc + n;
#pragma unroll
      for (int i = 0; i < CNT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = 4 * i + j;
          if (n + c < epi.N) {
            float t = v[i][j];
            if (epi.addend) t += to_f32(epi.addend[row + c]);
            t = to_f32(from_f32<T>(t));   // the value the store writes
            if (epi.bnb_y && !(to_f32(epi.bnb_y[row + c]) > 0.f)) t = 0.f;
            s[c] += t;
            q[c] += t * to_f32(epi.bnb_x[row + c]);
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < CNT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = v[i][j];
          if (epi.bias && n + 4 * i + j < epi.N) t += epi.bias[n + 4 * i + j];
          t = to_f32(from_f32<T>(t));   // the value the store writes
          s[4 * i + j] += t;
          q[4 * i + j] += t * t;
        }
    }
  }
  // backward mode: q_raw = sum g' x  ->  sum g' xhat = rstd (q_raw - mean s), in f64 per partial sum (linear, so the partial
  // sums of all workgroups still add up to the total)
  template <typename Epi> __device__ __forceinline__ double q_out(const Epi& epi, int c, float sv, float qv) const {
    if constexpr (Epi::kBnb) return ((double)qv - (double)epi.bnb_mean[c] * (double)sv) * (double)epi.bnb_rstd[c];
    else return (double)qv;
  }
  template <typename Epi> __device__ __forceinline__ void flush(const Epi& epi, int n, int l15) {
    // all-reduce over the 16 lanes of a DPP row (= the row group): four rotate-and-add steps, one v_add_f32_dpp each
#pragma unroll
    for (int e = 0; e < 4 * CNT; ++e) {
      s[e] = row16_sum(s[e]);
      q[e] = row16_sum(q[e]);
    }
    // every lane of the row group now holds all 2 * 4*CNT totals: lane l15 takes column l15's, so that the wave issues
    // ONE atomic instruction over 64 (or 2 x 32) consecutive doubles instead of 8*CNT instructions with 4 live lanes
    double* dst = epi.stats + (size_t)(blockIdx.x % epi.stats_ncopy) * 2 * epi.N;
    if constexpr (CNT == 4) {
      float sv = s[0], qv = q[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) {
        sv = l15 == e ? s[e] : sv;
        qv = l15 == e ? q[e] : qv;
      }
      if (n + l15 < epi.N) {
        atomicAdd(dst + n + l15, (double)sv);
        atomicAdd(dst + epi.N + n + l15, q_out(epi, n + l15, sv, qv));
      }
    } else if constexpr (CNT == 2) {
      const int c = l15 & 7;
      float sv = s[0], qv = q[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) {
        sv = c == e ? s[e] : sv;
        qv = c == e ? q[e] : qv;
      }
      if (n + c < epi.N) {
        if (l15 < 8) atomicAdd(dst + n + c, (double)sv);
        else atomicAdd(dst + epi.N + n + c, q_out(epi, n + c, sv, qv));
      }
    } else {
      if (l15 != 0) return;
#pragma unroll
      for (int e = 0; e < 4 * CNT; ++e)
        if (n + e < epi.N) {
          atomicAdd(dst + n + e, (double)s[e]);
          atomicAdd(dst + epi.N + n + e, q_out(epi, n + e, s[e], q[e]));
        }
    }
  }
};


// ---------------------------------------------------------------------------------------------
// NT kernel.  AMODE 0: dense A[M,K] (row stride lda).  AMODE 1/2: A is an NHWC tensor gathered
// im2col-style according to ConvGeom (K = R*S*Cg, k = (r*S+s)*Cg + c).  AMODE 2 is the fast path
// (R*S <= 32 and, for dgrad, unit stride): every row precomputes one base offset and a tap-validity bit
// mask, and the (tap, channel) position advances incrementally -- no division or bounds arithmetic in
// the k-loop.  AMODE 1 is the fully general gather (strided dgrad).
// Epi: functor  void operator()(int m, int n, const f32x4& v)  -- v = C[m][n..n+3] in f32.
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int AMODE, typename Epi>
__device__ __forceinline__ void igemm_nt_body(const NtArgs& a, const ConvGeom& g, const Epi& epi) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int BK = 8 * VEC;
  constexpr int AI = BM / 32, BI = BN / 32;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  typedef typename Mma<T>::Frag Frag;

  __shared__ uint4 smem[8 * (BM + BN)];
  uint4* sA = smem;
  uint4* sB = smem + 8 * BM;

  const int tid = threadIdx.x;
  const int tiles_n = (a.N + BN - 1) / BN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kc = tid & 7, r0 = tid >> 3;

  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;

  long long a_off[AI];
  int a_h[AI], a_w[AI];
  unsigned a_mask[AI];
  bool a_ok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + r0 + 32 * i;
    a_ok[i] = m < a.M;
    a_mask[i] = 0;
    if (AMODE == 0) {
      a_off[i] = (long long)m * a.lda;
      a_h[i] = a_w[i] = 0;
    } else {
      const int wm = m % g.Wm;
      const int t = m / g.Wm;
      const int hm = t % g.Hm;
      const int ni = t / g.Hm;
      a_off[i] = (long long)ni * g.Hg * g.Wg * g.ldg;
      a_h[i] = hm;
      a_w[i] = wm;
      if (AMODE == 2) {
        // base = source pixel of tap (0,0); tap (r,s) adds +-(r*dh*Wg + s*dw)*ldg
        const int bh = g.mode == 1 ? hm * g.sh - g.ph : hm + g.ph;
        const int bw = g.mode == 1 ? wm * g.sw - g.pw : wm + g.pw;
        a_off[i] += ((long long)bh * g.Wg + bw) * g.ldg;
        if (a_ok[i]) {
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
  }
  long long b_off[BI];
  bool b_ok[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int n = n0 + r0 + 32 * i;
    b_ok[i] = n < a.N;
    b_off[i] = (long long)n * a.ldb;
  }

  uint4 ra[AI], rb[BI];
  // incremental (tap, channel) position of this thread's k-chunk (AMODE 2); calls come with k0 += BK
  int f_tap = 0, f_r = 0, f_s = 0, f_c = kc * VEC;
  if (AMODE == 2) {
    while (f_c >= g.Cg) {
      f_c -= g.Cg;
      ++f_tap;
      if (++f_s == g.S) { f_s = 0; ++f_r; }
    }
  }
  const int taps = g.R * g.S;
  auto load_tiles = [&](int k0) {
    const int k = k0 + kc * VEC;
    const bool kok = k < a.K;
    int r = 0, s = 0, c = k;
    if (AMODE == 1) {
      const int tap = k / g.Cg;
      c = k - tap * g.Cg;
      r = tap / g.S;
      s = tap - r * g.S;
    }
    if (AMODE == 2) {
      const int toff = (f_r * g.dh * g.Wg + f_s * g.dw) * g.ldg;
      const long long koff = (long long)(g.mode == 1 ? toff : -toff) + f_c;
      const bool tok = f_tap < taps;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (tok && ((a_mask[i] >> f_tap) & 1u)) v = ldg16(A + a_off[i] + koff);
        ra[i] = v;
      }
      f_c += BK;
      while (f_c >= g.Cg) {
        f_c -= g.Cg;
        ++f_tap;
        if (++f_s == g.S) { f_s = 0; ++f_r; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (a_ok[i] && kok) {
          if (AMODE == 0) {
            v = ldg16(A + a_off[i] + k);
          } else {
            int hi, wi;
            if (conv_src(g, a_h[i], a_w[i], r, s, hi, wi))
              v = ldg16(A + a_off[i] + ((long long)hi * g.Wg + wi) * g.ldg + c);
          }
        }
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b_ok[i] && kok) v = ldg16(B + b_off[i] + k);
      rb[i] = v;
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm_ = wave & 1, wn_ = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // float32 operands: the MFMA accumulator is ONE round-to-nearest f32 chain over the whole reduction, whose error grows
  // like sqrt(K) -- 2.4x (K = 256) to 9x (K = 36864) the error of PyTorch's blocked CPU convolution against float64
  // (tools/diag_f32_error.py, profiles/r05_f32_error_by_layer.txt), the "constant 1.5x" of the Res50-PPM stage table.  The
  // f32 instantiation therefore keeps the chain one k-step (8 MFMA accumulations) long and carries the running total in
  // float64 registers; the total is rounded to f32 once, in the epilogue.  (bf16 operands: input rounding dominates by four
  // orders of magnitude; that path is unchanged.)
  constexpr bool WIDE = sizeof(T) == 4;
  double tot[WIDE ? TN : 1][WIDE ? TM : 1][4];
  if constexpr (WIDE) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[i][j][e] = 0.0;
  }

  load_tiles(0);
  for (int k0 = 0; k0 < a.K; k0 += BK) {
#pragma unroll
    for (int i = 0; i < AI; ++i) sA[kc * BM + ((r0 + 32 * i) ^ kc)] = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) sB[kc * BN + ((r0 + 32 * i) ^ kc)] = rb[i];
    __syncthreads();
    if (k0 + BK < a.K) load_tiles(k0 + BK);  // global loads stay in flight under the MFMAs
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
        for (int j = 0; j < TM; ++j) Mma<T>::run(acc[i][j], fb[i], fa[j]);  // D[n][m]
    }
    if constexpr (WIDE) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) tot[i][j][e] += (double)acc[i][j][e];
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
  }

  // row block outer, column block inner: the TN stores of one output row land back to back, so its 128-byte line is
  // completed in L2 before it can be evicted half-written (the 256x256 kernel wrote 2.5x its output bytes to HBM
  // with the loops the other way round)
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int m = m0 + wm_ * WTM + j * 16 + l15;
      const int n = n0 + wn_ * WTN + i * 16 + lg * 4;
      if constexpr (WIDE)
        epi(m, n, (f32x4){(float)tot[i][j][0], (float)tot[i][j][1], (float)tot[i][j][2], (float)tot[i][j][3]});
      else
        epi(m, n, acc[i][j]);
    }
}

// ---------------------------------------------------------------------------------------------
// Latency-optimised NT body for small, K-deep GEMMs on a dependency chain (LSTM steps):
// one BM x 64 output tile per workgroup (BM = 16 keeps >= 128 workgroups busy on the fused epilogue),
// the 4 waves split K four ways (each k-iteration stages BM x 4*BK of A and 64 x 4*BK of B), partial tiles are
// summed through LDS and the epilogue runs on the reduced f32 values.  4x fewer dependent
// load->MFMA round trips than igemm_nt_body for the same K.
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, typename Epi>
__device__ __forceinline__ void igemm_nt_ksplit_body(const NtArgs& a, const Epi& epi) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int BK = 8 * VEC;
  constexpr int TM = BM / 16;       // MFMA tiles along the (batch) row dimension per wave
  constexpr int AI = BM / 8;        // A-panel vectors per thread per k-iteration
  typedef typename Mma<T>::Frag Frag;
  // A panels [4 waves][8 chunks][BM rows], B panels [4][8][64]; reused as the f32 reduction buffer
  constexpr int A_VECS = 4 * 8 * BM, B_VECS = 4 * 8 * 64;
  constexpr int RED_VECS = 4 * BM * 64 / 4;
  __shared__ uint4 smem[(A_VECS + B_VECS) > RED_VECS ? (A_VECS + B_VECS) : RED_VECS];
  uint4* sA = smem;
  uint4* sB = smem + A_VECS;

  const int tid = threadIdx.x;
  const int tiles_n = (a.N + 63) / 64;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * 64;
  const int q = tid & 31;        // 16-byte chunk inside the 4*BK panel
  const int ws = q >> 3, kc = q & 7;
  const int r0 = tid >> 5;       // rows r0 + 8*i
  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;

  uint4 ra[AI], rb[8];
  auto load_panels = [&](int k0) {
    const int k = k0 + q * VEC;
    const bool kok = k < a.K;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int m = m0 + r0 + 8 * i;
      ra[i] = (kok && m < a.M) ? ldg16(A + (long long)m * a.lda + k) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + r0 + 8 * i;
      rb[i] = (kok && n < a.N) ? ldg16(B + (long long)n * a.ldb + k) : make_uint4(0, 0, 0, 0);
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  f32x4 acc[4][TM];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // epilogue operands of this thread's output vectors (epilogue element gidx = tid + 256*i -> (m, n)): in flight
  // during the whole k-loop
  typename Epi::Pre pre[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int gidx = tid + 256 * i;
    pre[i] = epi.prefetch(m0 + (gidx >> 4), n0 + (gidx & 15) * 4);
  }

  load_panels(0);
  for (int k0 = 0; k0 < a.K; k0 += 4 * BK) {
#pragma unroll
    for (int i = 0; i < AI; ++i) sA[ws * 8 * BM + kc * BM + ((r0 + 8 * i) ^ kc)] = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) sB[ws * 512 + kc * 64 + ((r0 + 8 * i) ^ kc)] = rb[i];
    __syncthreads();
    if (k0 + 4 * BK < a.K) load_panels(k0 + 4 * BK);
    if (k0 + wave * BK < a.K) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kcr = ks * 4 + lg;
        Frag fa[TM], fb[4];
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[j] = *(const Frag*)&sA[wave * 8 * BM + kcr * BM + ((j * 16 + l15) ^ kcr)];
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *(const Frag*)&sB[wave * 512 + kcr * 64 + ((i * 16 + l15) ^ kcr)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) Mma<T>::run(acc[i][j], fb[i], fa[j]);
      }
    }
    __syncthreads();
  }

  // cross-wave reduction through LDS: red[w][m][n] f32
  float* red = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = j * 16 + l15, n = i * 16 + lg * 4;
      *(f32x4*)&red[wave * BM * 64 + m * 64 + n] = acc[i][j];
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int gidx = tid + 256 * i;
    const int m = gidx >> 4, n = (gidx & 15) * 4;
    f32x4 v = *(const f32x4*)&red[m * 64 + n];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 u = *(const f32x4*)&red[w * BM * 64 + m * 64 + n];
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    epi.apply(m0 + m, n0 + n, v, pre[i]);
  }
}

// ---------------------------------------------------------------------------------------------
// Latency-optimised NT body for the LSTM recurrence (one BM x BN output tile per workgroup, K split over the
// 4 waves).  Every wave fetches its whole K slice as MFMA fragments straight from global memory (L2-resident
// operands: h_{t-1} / dgates and the recurrent weights) with all loads in flight at once -- no LDS staging, no
// barrier in the k-loop -- so a step costs ONE memory round trip + the MFMAs + the cross-wave reduction, instead
// of K/(4*BK) staged load->LDS->MFMA rounds.  The 16-byte vector a lane loads for (row l15, chunk lg) is exactly
// its fragment of one Mma<T>::run; A and B use the same k order so any K (multiple of the vector width) works.
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, typename Epi>
__device__ __forceinline__ void igemm_nt_kdirect_body(const NtArgs& a, const Epi& epi) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int KR = 4 * VEC;  // k extent of one Mma<T>::run
  constexpr int TM = BM / 16, TN = BN / 16, KC = 8;
  typedef typename Mma<T>::Frag Frag;
  __shared__ float red[4 * BM * BN];

  const int tid = threadIdx.x;
  const int tiles_n = (a.N + BN - 1) / BN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lg = lane >> 4;
  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int NV = BM * BN / 4;  // f32x4 vectors of the tile; thread tid owns vector tid (NV <= 256)
  static_assert(NV <= 256, "one epilogue vector per thread");
  const int em = tid / (BN / 4), en = (tid % (BN / 4)) * 4;
  typename Epi::Pre pre = epi.prefetch(tid < NV ? m0 + em : a.M, n0 + en);  // in flight during the k-loop

  const int kw = ((a.K + 4 * KR - 1) / (4 * KR)) * KR;  // per-wave K slice
  const int k_lo = wave * kw;
  const int k_hi = (k_lo + kw < a.K) ? k_lo + kw : a.K;
  for (int kp = k_lo; kp < k_hi; kp += KC * KR) {
    uint4 fa[TM][KC], fb[TN][KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int k = kp + c * KR + lg * VEC;
      const bool ok = k < k_hi;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int m = m0 + 16 * j + l15;
        fa[j][c] = (ok && m < a.M) ? ldg16(A + (long long)m * a.lda + k) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n0 + 16 * i + l15;
        fb[i][c] = (ok && n < a.N) ? ldg16(B + (long long)n * a.ldb + k) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          Mma<T>::run(acc[i][j], *(const Frag*)&fb[i][c], *(const Frag*)&fa[j][c]);
  }

  // cross-wave reduction through LDS: red[w][m][n] f32
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = j * 16 + l15, n = i * 16 + lg * 4;
      *(f32x4*)&red[wave * BM * BN + m * BN + n] = acc[i][j];
    }
  __syncthreads();
  if (tid < NV) {
    f32x4 v = *(const f32x4*)&red[em * BN + en];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 u = *(const f32x4*)&red[w * BM * BN + em * BN + en];
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    epi.apply(m0 + em, n0 + en, v, pre);
  }
}

template <typename T, int BM, int BN, int AMODE, typename Epi>
__global__ __launch_bounds__(256) void igemm_nt_kernel(NtArgs a, ConvGeom g, Epi epi) {
  igemm_nt_body<T, BM, BN, AMODE, Epi>(a, g, epi);
}

// ---------------------------------------------------------------------------------------------
// NT kernel v2: direct-to-LDS staging (global_load_lds, 16 B per lane), two LDS buffers, ONE barrier
// per k-step.  The loads for k-step t+1 are issued before the MFMAs of k-step t and land while they run;
// no staging VGPRs and no ds_write pass.
//
// LDS image per operand tile: [ROWS][8 chunks x 16 B] (128-byte rows); physical chunk = kc ^ ((row>>1)&7).
// A wave's LDS-DMA instruction writes 64 consecutive 16-byte slots = 8 full rows, so the image is
// lane-linear as the hardware requires and the swizzle lives in the per-lane SOURCE address: lane l of
// row group j fetches row 8j + (l>>3), logical chunk (l&7) ^ ((row>>1)&7) -- each row is still fetched
// as one full 128-byte line.  ds_read_b128 of an MFMA fragment (16 rows, one logical chunk) then hits
// 16 distinct 16-byte bank slots.  Padded taps / rows beyond M / k beyond K read a global zero page.
// AMODE 0: dense A.  AMODE 2: conv gather with per-row tap masks (R*S <= 32; dgrad: unit stride).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// branch-free select between a real source address and the zero page
__device__ __forceinline__ const void* sel_ptr(bool ok, const void* p, const void* z) {
  const unsigned long long m = 0ull - (unsigned long long)ok;
  return (const void*)(((unsigned long long)p & m) | ((unsigned long long)z & ~m));
}

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)gptr, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

// Same LDS-DMA through a raw buffer resource (buffer_load_dwordx4 ... offen lds): the address is a scalar
// descriptor + a 32-bit per-lane BYTE offset, and an offset beyond num_records returns zeros -- so a padded /
// out-of-range vector costs one v_cndmask (offset := 0xFFFFFFFF) instead of a 64-bit pointer add plus a 64-bit select
// against the zero page.  num_records = 2 GiB: operands must be smaller than that (the host checks).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, (int)0x80000000u, 0x00020000);
}
__device__ __forceinline__ void glds16_buf(rsrc_t r, bool ok, int elem_off, int esize_shift, void* lds_wave_base) {
  const unsigned voff = ok ? ((unsigned)elem_off << esize_shift) : 0xFFFFFFFFu;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

// AMODE 0: dense A.  AMODE 2: conv, Cg % BK == 0 (every k-step lies inside one tap: the tap state is scalar and
// advances incrementally).  AMODE 3: conv with small / odd Cg (first layer): per-vector tap arithmetic.
// NST = number of LDS stage buffers.  2 (default): one k-step of prefetch, __syncthreads() per k-step -- right when several
// workgroups share a CU and hide each other's load latency.  NST > 2 (dynamic LDS, NST * (BM + BN) * 128 bytes): NST - 1
// k-steps of LDS-DMA stay in flight ACROSS the barriers (raw s_barrier + counted s_waitcnt vmcnt; __syncthreads() would drain
// them) -- for launches with at most ~one workgroup per CU (small-M layers: batch-2 detector, batch-32 recogniser), whose
// k-step time is otherwise one full L2 / HBM round trip.
template <typename T, int BM, int BN, int AMODE, typename Epi, int NST = 2>
__global__ __launch_bounds__(256) void igemm_nt_glds_kernel(NtArgs a, ConvGeom g, Epi epi) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int BK = 8 * VEC;
  constexpr int AI = BM / 32, BI = BN / 32;  // 8-row groups per wave
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  constexpr int TILE_VECS = (BM + BN) * 8;
  typedef typename Mma<T>::Frag Frag;

  __shared__ uint4 smem_static[NST == 2 ? 2 * TILE_VECS : 1];
  extern __shared__ uint4 smem_dynamic[];
  uint4* const smem = NST == 2 ? smem_static : smem_dynamic;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M - a.m_begin + BM - 1) / BM;
  // XCD-aware tile map (grid = ceil(tiles_m/8)*8*tiles_n): hardware workgroup id b runs on XCD b % 8; the column
  // tiles of one row tile -- which gather the same activation rows -- take consecutive slots of the SAME XCD, so
  // the rows are fetched into one L2 instead of tiles_n of them (measured on the N = 512 layers: FETCH_SIZE per
  // launch was ~5x the algorithmic bytes with the plain b -> (b / tiles_n, b % tiles_n) map).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_m = (slot / tiles_n) * 8 + xcd, tile_n = slot % tiles_n;
  if (tile_m >= tiles_m) return;
  const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;
  const int lrow = lane >> 3, lpc = lane & 7;

  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;
  (void)a.zero;

  // ---- per-thread row descriptors: source pointer of (row, logical chunk) at k = 0 / tap (0,0)
  // operands are staged through raw buffer resources: 32-bit element offsets, out-of-range -> zeros (glds16_buf)
  constexpr int ESH = sizeof(T) == 2 ? 1 : 2;
  const rsrc_t rsA = make_rsrc(A), rsB = make_rsrc(B);
  int a_ptr[AI];
  unsigned a_mask[AI];
  int a_kc[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (wave * AI + i) * 8 + lrow;
    const int m = m0 + row;
    a_kc[i] = (lpc ^ ((row >> 1) & 7)) * VEC;
    a_mask[i] = 0;
    a_ptr[i] = 0;
    if (AMODE == 0) {
      a_ptr[i] = (int)((long long)m * a.lda + a_kc[i]);
      a_mask[i] = m < a.M ? 1u : 0u;
    } else if (m < a.M) {
      const int wm = m % g.Wm;
      const int t = m / g.Wm;
      const int hm = t % g.Hm;
      const int ni = t / g.Hm;
      const int bh = g.mode == 1 ? hm * g.sh - g.ph : hm + g.ph;
      const int bw = g.mode == 1 ? wm * g.sw - g.pw : wm + g.pw;
      a_ptr[i] = (int)((long long)ni * g.Hg * g.Wg * g.ldg + ((long long)bh * g.Wg + bw) * g.ldg +
                       (AMODE == 2 ? a_kc[i] : 0));
      unsigned msk = 0;
      for (int r = 0; r < g.R; ++r)
        for (int s2 = 0; s2 < g.S; ++s2) {
          int hi, wi;
          if (conv_src(g, hm, wm, r, s2, hi, wi)) msk |= 1u << (r * g.S + s2);
        }
      a_mask[i] = msk;
    }
  }
  int b_ptr[BI];
  int b_kc[BI];
  bool b_ok[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = (wave * BI + i) * 8 + lrow;
    // LDS row -> output channel: inside each wave tile (WTN columns) LDS row i16*16 + lg*4 + r holds channel
    // lg*(4*TN) + i16*4 + r, so that the 4 channels a lane gets from each of its TN column blocks are consecutive
    // (TN*4-channel runs per lane -> 16-byte stores in the epilogue)
    const int rb = row % WTN;
    const int n = n0 + (row - rb) + ((rb >> 2) & 3) * (4 * TN) + (rb >> 4) * 4 + (rb & 3);
    b_kc[i] = (lpc ^ ((row >> 1) & 7)) * VEC;
    b_ok[i] = n < a.N;
    b_ptr[i] = (int)((long long)n * a.ldb + b_kc[i]);
  }

  const int taps = g.R * g.S;
  const int sgn = g.mode == 1 ? 1 : -1;
  const bool k_exact = (a.K % BK) == 0;  // no partial last k-step: skip the k < K predicate
  // scalar tap state of the NEXT k-step to stage (AMODE 2)
  int s_tap = 0, s_r = 0, s_s = 0, s_c0 = 0;

  auto stage = [&](uint4* sA, int k0) {
    uint4* sB = sA + BM * 8;
    if (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const bool ok = a_mask[i] && (k_exact || k0 + a_kc[i] < a.K);
        glds16_buf(rsA, ok, a_ptr[i] + k0, ESH, sA + (wave * AI + i) * 64);
      }
    } else if (AMODE == 2) {
      const int koff = sgn * ((s_r * g.dh * g.Wg + s_s * g.dw) * g.ldg) + s_c0;
      const unsigned bit = 1u << s_tap;
#pragma unroll
      for (int i = 0; i < AI; ++i) glds16_buf(rsA, a_mask[i] & bit, a_ptr[i] + koff, ESH, sA + (wave * AI + i) * 64);
      s_c0 += BK;
      if (s_c0 >= g.Cg) {
        s_c0 = 0;
        ++s_tap;
        if (++s_s == g.S) { s_s = 0; ++s_r; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int k = k0 + a_kc[i];
        const int tap = k / g.Cg;
        const int c = k - tap * g.Cg;
        const int r = tap / g.S, s2 = tap - r * g.S;
        const bool ok = tap < taps && ((a_mask[i] >> tap) & 1u);
        glds16_buf(rsA, ok, a_ptr[i] + sgn * ((r * g.dh * g.Wg + s2 * g.dw) * g.ldg) + c, ESH,
                   sA + (wave * AI + i) * 64);
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const bool ok = b_ok[i] && (k_exact || k0 + b_kc[i] < a.K);
      glds16_buf(rsB, ok, b_ptr[i] + k0, ESH, sB + (wave * BI + i) * 64);
    }
  };

  const int wm_ = wave & 1, wn_ = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;
  // fragment read addresses (vector index inside one stage): row*8 + (kc ^ ((row>>1)&7)), kc = ks*4 + lg.
  // (row>>1)&7 == l15>>1 for every tile row of this lane, so kc^x = (lg^x) ^ (ks*4): two bases per row.
  const int xsw = (l15 >> 1) & 7;
  int fa_off[TM][2], fb_off[TN][2];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int r = wm_ * WTM + j * 16 + l15;
    fa_off[j][0] = r * 8 + (lg ^ xsw);
    fa_off[j][1] = r * 8 + ((lg ^ xsw) ^ 4);
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int r = wn_ * WTN + i * 16 + l15;
    fb_off[i][0] = BM * 8 + r * 8 + (lg ^ xsw);
    fb_off[i][1] = BM * 8 + r * 8 + ((lg ^ xsw) ^ 4);
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // float32 operands: f32 chains of one k-step, running total in float64 (see igemm_nt_body)
  constexpr bool WIDE = sizeof(T) == 4;
  double tot[WIDE ? TN : 1][WIDE ? TM : 1][4];
  if constexpr (WIDE) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[i][j][e] = 0.0;
  }

  auto compute = [&](const uint4* st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag fa[TM], fb[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[j] = *(const Frag*)&st[fa_off[j][ks]];
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[i] = *(const Frag*)&st[fb_off[i][ks]];
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::run(acc[i][j], fb[i], fa[j]);
    }
    if constexpr (WIDE) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) tot[i][j][e] += (double)acc[i][j][e];
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
  };

  // this workgroup's share of the reduction: k-steps [split * per, split * per + nk) of nk_all (NtArgs.ksplit; 1: all of them)
  const int nk_all = (a.K + BK - 1) / BK;
  int nk = nk_all, kb = 0;
  if (a.ksplit > 1) {
    const int per = (nk_all + a.ksplit - 1) / a.ksplit;
    const int first = (int)blockIdx.y * per;
    kb = first * BK;
    nk = min(per, nk_all - first);     // may be <= 0 for the last splits: they contribute a zero slab
    if (AMODE == 2) {
      s_tap = kb / g.Cg;
      s_c0 = kb - s_tap * g.Cg;
      s_r = s_tap / g.S;
      s_s = s_tap - s_r * g.S;
    }
  }
  if constexpr (NST == 2) {
    uint4* st0 = smem;
    uint4* st1 = smem + TILE_VECS;
    if (nk > 0) stage(st0, kb);
    // two k-steps per iteration so that the stage base is a compile-time constant in every LDS access
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      __syncthreads();  // k-step t landed (vmcnt drained before the barrier); stage 1 is free
      stage(st1, kb + (t + 1) * BK);
      compute(st0);
      __syncthreads();
      if (t + 2 < nk) stage(st0, kb + (t + 2) * BK);
      compute(st1);
    }
    if (t < nk) {
      __syncthreads();
      compute(st0);
    }
  } else {
    // NST buffers, NST - 1 k-steps in flight.  Iteration t: wait until this wave's part of k-step t has landed (counted vmcnt:
    // the younger stages stay in flight), raw barrier (everybody's part landed; everybody is done reading buffer (t-1) % NST),
    // refill that buffer with k-step t + NST - 1, compute on buffer t % NST.  NST k-steps per trip of the outer loop keep every
    // LDS base a compile-time constant.
    constexpr int LPS = AI + BI;   // LDS-DMA instructions per stage and wave
    static_assert((NST - 2) * LPS <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
    for (int u = 0; u < NST - 1; ++u)
      if (u < nk) stage(smem + u * TILE_VECS, kb + u * BK);
    for (int t0 = 0; t0 < nk; t0 += NST) {
#pragma unroll
      for (int u = 0; u < NST; ++u) {
        const int t = t0 + u;
        if (t < nk) {
          const int younger = min(NST - 2, nk - 1 - t);   // stages issued after k-step t that may stay in flight
          if (younger >= 2 && NST >= 4) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
          else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPS) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
          if (t + NST - 1 < nk) stage(smem + ((u + NST - 1) % NST) * TILE_VECS, kb + (t + NST - 1) * BK);
          compute(smem + u * TILE_VECS);
        }
      }
    }
  }

  if constexpr (!WIDE) {
    if (a.ksplit > 1) {
      // partial tile -> own slab (register order: element (k, tid) = accumulator tile k of thread tid), ticket, and the last
      // arriver of the tile sums all slabs in split order (the protocol of the TN kernels' group reduction: sc1 stores, the
      // ticket behind s_waitcnt vmcnt(0) + barrier, sc1 loads)
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
      const int tile_lin = tile_m * tiles_n + tile_n;
      const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char*)a.ws + NT_SPLIT_TICKETS * 4, (short)0, 0x7fffffff, 0x00020000);
      constexpr int SLAB = BM * BN * 4;
      const int so = (tile_lin * a.ksplit + (int)blockIdx.y) * SLAB;
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, acc[i][j]), rs, tid * 16, so + (i * TM + j) * 4096, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();       // (also: every wave is done with the stage buffers)
      volatile int* bc = (volatile int*)smem;
      int* tk = (int*)a.ws + (tile_lin % NT_SPLIT_TICKETS);
      if (tid == 0) *bc = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (*bc != a.ksplit - 1) return;
      if (tid == 0) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int sp = 0; sp < a.ksplit; ++sp) {
        const int sj = (tile_lin * a.ksplit + sp) * SLAB;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, sj + (i * TM + j) * 4096, 16));
      }
    }
  }

  // row block outer, column block inner: the TN stores of one output row land back to back, so its 128-byte line is
  // completed in L2 before it can be evicted half-written (the 256x256 kernel wrote 2.5x its output bytes to HBM
  // with the loops the other way round)
  EpiColStats<T, TN> cst;
  bool with_stats = false;
  if constexpr (EpiHasStats<Epi>::value) with_stats = epi.stats != nullptr;
  if (with_stats) cst.init();
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    f32x4 run[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if constexpr (WIDE) run[i] = (f32x4){(float)tot[i][j][0], (float)tot[i][j][1], (float)tot[i][j][2], (float)tot[i][j][3]};
      else run[i] = acc[i][j];
    }
    if constexpr (EpiHasStats<Epi>::value)
      if (with_stats) cst.add(epi, m0 + wm_ * WTM + j * 16 + l15, n0 + wn_ * WTN + lg * (4 * TN), run);
    epi.template store_run<TN>(m0 + wm_ * WTM + j * 16 + l15, n0 + wn_ * WTN + lg * (4 * TN), run);
  }
  if constexpr (EpiHasStats<Epi>::value)
    if (with_stats) cst.flush(epi, n0 + wn_ * WTN + lg * (4 * TN), l15);
}

// ---------------------------------------------------------------------------------------------
// NT kernel v3 ("big tile"): the same direct-to-LDS / XOR-swizzled scheme as igemm_nt_glds_kernel with
// WM x WN waves (8 waves = 512 threads) and a (WM*TM*16) x (WN*TN*16) tile, e.g. 256x256 or 288x256.
// Why: with a 64x64 wave tile the v2 kernel moves one LDS byte per 32 flops and its prefetch distance (one k-step =
// 32 MFMAs per wave = ~210 ns) is shorter than the L2/HBM latency, so it plateaus at ~1/3 of the MFMA peak.  A
// 128x64 (or 144x64) wave tile needs 0.75x the ds_read bytes and half the LDS-DMA bytes per flop, and one k-step
// of MFMAs (64-72 per wave, two waves per SIMD) covers a full memory round trip.
// LDS: 2 stages x (BM + BN) x 128 B (128-139 KB, dynamic) -> one workgroup per CU; the host picks this kernel
// only when the tile count fills the 256 CUs in whole rounds (see nt_big_choice in gemm_conv.hip).
// Tile -> workgroup map is XCD-aware: the column tiles of one row tile (which re-read the same activation rows)
// get consecutive slots of the SAME XCD (hardware workgroup id % 8 = XCD), so the re-read hits that XCD's L2.
// ---------------------------------------------------------------------------------------------
template <typename T, int WM, int WN, int TM, int TN, int AMODE, typename Epi>
__global__ __launch_bounds__(64 * WM * WN) void igemm_nt_big_kernel(NtArgs a, ConvGeom g, Epi epi) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int BK = 8 * VEC;
  constexpr int NW = WM * WN;
  constexpr int WTM = TM * 16, WTN = TN * 16, BM = WM * WTM, BN = WN * WTN;
  constexpr int AG = BM / 8, BG = BN / 8;                      // 8-row staging groups
  constexpr int AI = (AG + NW - 1) / NW, BI = (BG + NW - 1) / NW;
  constexpr int TILE_VECS = (BM + BN) * 8;
  typedef typename Mma<T>::Frag Frag;
  static_assert(AMODE == 0 || AMODE == 2, "big-tile kernel: dense or fast conv gather only");

  extern __shared__ uint4 smem_big[];
  uint4* smem = smem_big;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int trows = a.tile_rows > 0 ? a.tile_rows : BM;     // rows of M a tile covers (<= BM)
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M - a.m_begin + trows - 1) / trows;
  // XCD-aware map: hardware block b -> xcd = b & 7, slot = b >> 3
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_m = (slot / tiles_n) * 8 + xcd, tile_n = slot % tiles_n;
  if (tile_m >= tiles_m) return;
  const int m0 = a.m_begin + tile_m * trows, n0 = tile_n * BN;
  const int lrow = lane >> 3, lpc = lane & 7;

  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;
  (void)a.zero;
  constexpr int ESH = sizeof(T) == 2 ? 1 : 2;
  const rsrc_t rsA = make_rsrc(A), rsB = make_rsrc(B);

  // per-thread row descriptors: 32-bit element offset of (row, logical chunk) at k = 0 / tap (0,0) + tap mask.
  // (operands are < 2^31 elements; offsets instead of pointers halve the descriptor registers)
  auto kc_of = [&](int gi) { return (lpc ^ (((gi * 8 + lrow) >> 1) & 7)) * VEC; };
  int a_off[AI];
  unsigned a_mask[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int gi = wave + i * NW;
    const int m = m0 + gi * 8 + lrow;
    a_mask[i] = 0;
    a_off[i] = 0;
    if (gi < AG && gi * 8 + lrow < trows) {
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
    const int rb = row % WTN;  // LDS row -> channel permutation inside a wave tile, see igemm_nt_glds_kernel
    const int n = n0 + (row - rb) + ((rb >> 2) & 3) * (4 * TN) + (rb >> 4) * 4 + (rb & 3);
    if (gi < BG && n < a.N) b_okmask |= 1u << i;
    b_off[i] = (int)((long long)n * a.ldb + kc_of(gi));
  }

  const int sgn = g.mode == 1 ? 1 : -1;
  const bool k_exact = (a.K % BK) == 0;
  int s_tap = 0, s_r = 0, s_s = 0, s_c0 = 0;  // scalar tap state of the NEXT k-step to stage (AMODE 2)

  auto stage = [&](uint4* sA, int k0) {
    uint4* sB = sA + BM * 8;
    if (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int gi = wave + i * NW;
        if (AG % NW == 0 || gi < AG) {
          const bool ok = a_mask[i] && (k_exact || k0 + kc_of(gi) < a.K);
          glds16_buf(rsA, ok, a_off[i] + k0, ESH, sA + gi * 64);
        }
      }
    } else {
      const int koff = sgn * ((s_r * g.dh * g.Wg + s_s * g.dw) * g.ldg) + s_c0;
      const unsigned bit = 1u << s_tap;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int gi = wave + i * NW;
        if (AG % NW == 0 || gi < AG) glds16_buf(rsA, a_mask[i] & bit, a_off[i] + koff, ESH, sA + gi * 64);
      }
      s_c0 += BK;
      if (s_c0 >= g.Cg) {
        s_c0 = 0;
        ++s_tap;
        if (++s_s == g.S) { s_s = 0; ++s_r; }
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int gi = wave + i * NW;
      if (BG % NW == 0 || gi < BG) {
        const bool ok = ((b_okmask >> i) & 1u) && (k_exact || k0 + kc_of(gi) < a.K);
        glds16_buf(rsB, ok, b_off[i] + k0, ESH, sB + gi * 64);
      }
    }
  };

  const int wm_ = wave % WM, wn_ = wave / WM;
  const int l15 = lane & 15, lg = lane >> 4;
  const int xsw = (l15 >> 1) & 7;
  // fragment base offsets (vector index inside a stage) of this lane: row*8 + (kc ^ xsw), kc = ks*4 + lg
  const int fa_base = (wm_ * WTM + l15) * 8 + (lg ^ xsw);
  const int fb_base = BM * 8 + (wn_ * WTN + l15) * 8 + (lg ^ xsw);

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const uint4* st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4* sa = st + (ks ? (fa_base ^ 4) : fa_base);
      const uint4* sb = st + (ks ? (fb_base ^ 4) : fb_base);
      Frag fb[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[i] = *(const Frag*)&sb[i * 128];
      // A fragments are streamed (two in flight) rather than all TM held at once: keeps the 288-row variant inside
      // the 256-register budget of two waves per SIMD
      Frag fa = *(const Frag*)&sa[0];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        Frag nxt = fa;
        if (j + 1 < TM) nxt = *(const Frag*)&sa[(j + 1) * 128];
#pragma unroll
        for (int i = 0; i < TN; ++i) Mma<T>::run(acc[i][j], fb[i], fa);
        fa = nxt;
      }
    }
  };

  const int nk = (a.K + BK - 1) / BK;
  uint4* st0 = smem;
  uint4* st1 = smem + TILE_VECS;
  if (nk > 0) stage(st0, 0);
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    __syncthreads();  // k-step t landed (vmcnt drained before the barrier); stage 1 is free
    stage(st1, (t + 1) * BK);
    compute(st0);
    __syncthreads();
    if (t + 2 < nk) stage(st0, (t + 2) * BK);
    compute(st1);
  }
  if (t < nk) {
    __syncthreads();
    compute(st0);
  }

  if constexpr (EpiIsPool<Epi>::value) {
    // conv + bias + ReLU + MAX-POOL in one launch (EpiPool below): the tile goes to LDS as T [BM][BN] (16-byte chunk c of row r
    // at chunk c ^ (r & (CH-1)): conflict-free both ways), then every thread pools whole channel vectors out of it with the
    // comparison rule of maxpool_fwd_kernel (first maximum in window order, NaN wins) -- same values, same arg-max codes.
    static_assert(sizeof(T) == 2, "pooled epilogue: bf16 only");
    constexpr int CH = BN / 8;
    static_assert((CH & (CH - 1)) == 0 && (4 * TN) % 8 == 0, "pooled epilogue: tile width / lane run in whole 16-byte chunks");
    __syncthreads();                       // every wave is done reading the stage buffers (no LDS-DMA is in flight here)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int r = wm_ * WTM + j * 16 + l15;
#pragma unroll
      for (int q = 0; q < TN / 2; ++q) {   // one 16-byte chunk = the 8 channels of accumulator tiles 2q, 2q+1
        const int n = n0 + wn_ * WTN + lg * (4 * TN) + q * 8;
        T o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = acc[2 * q + e / 4][j][e % 4];
          if (epi.bias && n + e < epi.N) t += epi.bias[n + e];
          if (epi.relu) t = fmaxf(t, 0.f);
          o[e] = from_f32<T>(t);
        }
        const int c = (wn_ * WTN + lg * (4 * TN)) / 8 + q;
        smem[r * CH + (c ^ (r & (CH - 1)))] = *(const uint4*)o;
      }
    }
    __syncthreads();
    // pooled elements of this tile: thread -> channel chunk c = tid % CH (fixed) and pooled pixels tid / CH, + 64 NW / CH, ...
    // in (pooled row, pooled column) order; the indices advance incrementally (no division in the loop)
    const int img_rows = trows / epi.Wo;                 // conv image-rows the tile covers (a multiple of kh)
    const int prows = img_rows / epi.kh;
    const int row0 = m0 / epi.Wo;                        // global conv image-row of the tile's first row (m0 % Wo == 0)
    constexpr int PSTEP = 64 * NW / CH;                  // pooled pixels per sweep of the workgroup
    const int c = tid % CH;
    int pix = tid / CH;
    int pr = pix / epi.PWo, pwi = pix - pr * epi.PWo;
    int nimg = (row0 + pr * epi.kh) / epi.Ho, h = (row0 + pr * epi.kh) - nimg * epi.Ho;
    const bool col_ok = n0 + c * 8 < epi.N;
    auto pool_one = [&](auto kh_tag, auto kw_tag) {
      constexpr int KH = decltype(kh_tag)::value, KW = decltype(kw_tag)::value;   // 0 = runtime window
      const int kh = KH ? KH : epi.kh, kw = KW ? KW : epi.kw;
      for (; pr < prows; ) {
        if (nimg < epi.Nimg && col_ok) {
          float best[8];
          unsigned char bi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
          bool first = true;
#pragma unroll
          for (int i = 0; i < (KH ? KH : 1); ++i)
            for (int i2 = 0; i2 < (KH ? 1 : kh); ++i2) {
              const int ii = KH ? i : i2;
#pragma unroll
              for (int jx = 0; jx < (KW ? KW : 1); ++jx)
                for (int j2 = 0; j2 < (KW ? 1 : kw); ++j2) {
                  const int jj = KW ? jx : j2;
                  const int w = pwi * epi.sw - epi.pw + jj;
                  if ((unsigned)w >= (unsigned)epi.Wo) continue;
                  const int r = (pr * kh + ii) * epi.Wo + w;
                  const uint4 v = smem[r * CH + (c ^ (r & (CH - 1)))];
                  const T* pv = (const T*)&v;
#pragma unroll
                  for (int e = 0; e < 8; ++e) {
                    const float f = to_f32(pv[e]);
                    if (first || f > best[e] || f != f) { best[e] = f; bi[e] = (unsigned char)(ii * kw + jj); }
                  }
                  first = false;
                }
            }
          T o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(best[e]);
          const long long q = (((long long)nimg * epi.PHo + h / kh) * epi.PWo + pwi) * epi.ldc + n0 + c * 8;
          *(uint4*)(epi.C + q) = *(const uint4*)o;
          *(uint2*)(epi.idx + q) = *(const uint2*)bi;
        }
        pwi += PSTEP;
        while (pwi >= epi.PWo) {
          pwi -= epi.PWo;
          ++pr;
          h += kh;
          if (h >= epi.Ho) { h -= epi.Ho; ++nimg; }
        }
      }
    };
    if (epi.kh == 2 && epi.kw == 2) pool_one(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
    else pool_one(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    return;
  } else {
  // row block outer, column block inner: the TN stores of one output row land back to back, so its 128-byte line is
  // completed in L2 before it can be evicted half-written (the 256x256 kernel wrote 2.5x its output bytes to HBM
  // with the loops the other way round)
  EpiColStats<T, TN> cst;
  bool with_stats = false;
  if constexpr (EpiHasStats<Epi>::value) with_stats = epi.stats != nullptr;
  if (with_stats) cst.init();
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    f32x4 run[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) run[i] = acc[i][j];
    if constexpr (EpiHasStats<Epi>::value)
      if (with_stats) cst.add(epi, m0 + wm_ * WTM + j * 16 + l15, n0 + wn_ * WTN + lg * (4 * TN), run);
    epi.template store_run<TN>(m0 + wm_ * WTM + j * 16 + l15, n0 + wn_ * WTN + lg * (4 * TN), run);
  }
  if constexpr (EpiHasStats<Epi>::value)
    if (with_stats) cst.flush(epi, n0 + wn_ * WTN + lg * (4 * TN), l15);
  }
}

// Pooled epilogue of igemm_nt_big_kernel: y_pool = maxpool(relu(conv + bias)) and the arg-max codes of mr_maxpool_fwd, written
// directly -- the full-resolution activation never reaches HBM (round 6; reference backbones/crnn.py:14-33: Conv2d -> ReLU ->
// MaxPool2d).  Geometry contract (checked by the host, conv_pool_plan in gemm_conv.hip): non-overlapping window rows without
// vertical padding (kh == stride_h, pad_h == 0, Ho % kh == 0), NtArgs.tile_rows a multiple of kh * Wo that divides or is a
// multiple of Ho * Wo; any horizontal window (kw, sw, pw).  C / idx: [Nimg, PHo, PWo, ldc] in T / one byte per element.
template <typename T> struct EpiPool {
  T* C;
  unsigned char* idx;
  long long ldc;
  const float* bias;
  int relu;
  int M, N;
  int Nimg, Ho, Wo, kh, kw, sw, pw, PHo, PWo;
};
template <typename T> struct EpiIsPool<EpiPool<T>> { static constexpr bool value = true; };

// Plain epilogue: C = act(acc + bias [+ addend]) stored as T, row-major with leading dim ldc.
template <typename T> struct EpiStore {
  T* C;
  long long ldc;
  const float* bias;  // [N] or null
  int relu;
  int M, N;
  int vec_ok;  // ldc % 4 == 0 and C 16-byte aligned
  // optional BatchNorm batch statistics of the stored matrix (direct-to-LDS kernels only): stats[copy][0][n] += sum_m C[m][n],
  // stats[copy][1][n] += sum_m C[m][n]^2 over the values AS STORED (rounded to T); copy = workgroup % stats_ncopy
  double* stats = nullptr;
  int stats_ncopy = 1;
  // optional second summand with C's own layout (same ldc, 16-byte aligned when vec_ok): C = act(acc + bias + addend).  The
  // residual branch's gradient of a ResNet block rides in the epilogue of the dgrad of the block's first convolution instead of
  // an elementwise add kernel over the block input's gradient (mr_conv2d_dgrad_add; reference backbones/resnet.py:152-181).
  const T* addend = nullptr;
  static constexpr bool kBnb = false;   // see EpiStoreB
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
    if (m >= M || n >= N) return;
    if (bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < N) v[j] += bias[n + j];
    }
    if (addend) {
      const T* a = addend + (long long)m * ldc + n;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < N) v[j] += to_f32(a[j]);
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    T* p = C + (long long)m * ldc + n;
    if (vec_ok && n + 3 < N) {
      store4(p, v);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < N) p[j] = from_f32<T>(v[j]);
    }
  }
  // CNT*4 consecutive channels n .. n+4*CNT-1 of row m (the direct-to-LDS kernels permute the B rows inside each
  // wave tile so that a lane owns consecutive channels across its column blocks): 16-byte stores
  template <int CNT>
  __device__ __forceinline__ void store_run(int m, int n, const f32x4* v) const {
    constexpr int E16 = 16 / (int)sizeof(T);  // elements per 16-byte store
    if (m >= M) return;
    if (vec_ok && (ldc % E16) == 0 && (n % E16) == 0 && n + 4 * CNT <= N && (4 * CNT) % E16 == 0) {
      uint4* dst = (uint4*)(C + (long long)m * ldc + n);
      if (addend) {   // one 16-byte chunk at a time: at most 8 extra live registers next to the accumulators
        const uint4* asrc = (const uint4*)(addend + (long long)m * ldc + n);
#pragma unroll
        for (int q = 0; q < 4 * CNT / E16; ++q) {
          const uint4 av = asrc[q];
          const T* ap = (const T*)&av;
          T out[E16];
#pragma unroll
          for (int e = 0; e < E16; ++e) {
            const int c = q * E16 + e;
            float t = v[c / 4][c % 4] + to_f32(ap[e]);
            if (bias) t += bias[n + c];
            if (relu) t = fmaxf(t, 0.f);
            out[e] = from_f32<T>(t);
          }
          dst[q] = *(const uint4*)out;
        }
        return;
      }
      float x[4 * CNT];
#pragma unroll
      for (int i = 0; i < CNT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = v[i][j];
          if (bias) t += bias[n + 4 * i + j];
          if (relu) t = fmaxf(t, 0.f);
          x[4 * i + j] = t;
        }
      T out[4 * CNT];
#pragma unroll
      for (int e = 0; e < 4 * CNT; ++e) out[e] = from_f32<T>(x[e]);
      const uint4* src = (const uint4*)out;
#pragma unroll
      for (int q = 0; q < 4 * CNT / E16; ++q) dst[q] = src[q];
    } else {
#pragma unroll
      for (int i = 0; i < CNT; ++i) (*this)(m, n + 4 * i, v[i]);
    }
  }
};
// EpiStore whose statistics epilogue runs in BatchNorm-BACKWARD mode (mr_conv2d_dgrad_bnb): the stored matrix is the gradient
// dy of a training-mode BatchNorm's output y = act(bn(x) [+ residual]), C's own layout.  `stats` receives
//   stats[copy][0][n] += sum_m g'[m][n],   stats[copy][1][n] += sum_m g'[m][n] * xhat[m][n]
// (g' = dy where y > 0 if bnb_y is given, else dy; xhat = (x - mean) rstd) -- the accumulator layout of mr_bn_bwd, which then
// skips its reduction pass over dy / x / y.  A separate TYPE, instantiated for the 4-wave direct-to-LDS kernels only: as a
// runtime branch of EpiStore it pushed the 272x256 kernel (249 VGPRs) into 560 bytes of scratch per lane.
template <typename T> struct EpiStoreB : EpiStore<T> {
  const T* bnb_x = nullptr;
  const T* bnb_y = nullptr;
  const float* bnb_mean = nullptr;
  const float* bnb_rstd = nullptr;
  static constexpr bool kBnb = true;
};
#ifndef MR_NO_EPI_STATS   // (A/B build of tools/: the NT kernels compiled without the statistics epilogue)
template <typename T> struct EpiHasStats<EpiStore<T>> { static constexpr bool value = true; };
template <typename T> struct EpiHasStats<EpiStoreB<T>> { static constexpr bool value = true; };
#endif

// ---------------------------------------------------------------------------------------------
// TN kernel: C[NA,NB] (f32, atomically accumulated) += sum_{p in split} A[p,NA] * Bg[p,NB]
// BMODE 0: B dense [P, ldb].  BMODE 1: B = im2col(X) per ConvGeom (mode 1), NB = R*S*Cg.
// Tile 128 x 128, 4 waves of 64 x 64.  grid = (tiles_a*tiles_b, 1, splits).
// row_perm_h > 0: output row r is written to row  (r/ (4*h))*(4*h) + (r%4)*h + (r%(4*h))/4
//   (gate-interleaved LSTM rows back to PyTorch's i,f,g,o-major order).
// ---------------------------------------------------------------------------------------------
struct TnArgs {
  const void* A;
  const void* B;
  float* C;
  int P, NA, NB;
  long long lda, ldb;
  int ldc;
  int p_chunk;  // rows of P per split (multiple of the p-step)
  int row_perm_h;
  float* colsum;  // optional [NA]: += sum_p A[p, na] (bias gradient), accumulated by the tile_b == 0 blocks
  float* colsum2 = nullptr;  // optional second destination of the same column sums (an LSTM layer's b_ih and b_hh share theirs)
  const int2* rowtab;  // BMODE 2: per output pixel {element offset of its window's top-left input pixel, tap mask}
  // igemm_tn_glds_kernel only -- in-launch reduction of the split partials (see tn_taps.hip, TapArgs.grp): groups of
  // `grp` consecutive splits publish their 64 KB accumulator slabs (sc1) and take a ticket; the last arriver sums the
  // group's slabs and stores: plain read-modify-write when the group is ALL the splits of the tile, f32 atomics otherwise.
  // ws = [TN_TICKETS ints (zero between launches)][gridDim.x slabs x 65536 B]; grp <= 1: atomics only.
  int grp;
  void* ws;
  // fin == 2: no in-launch reduction -- every workgroup stores its 128 x 128 partial tile to its own 64 KB slab with plain
  // stores and tn_finalize_kernel (a second launch) adds the sum over the splits into C.  0: groups / atomics as above.
  int fin = 0;
};
constexpr int TN_TICKETS = 4096;

// Row table of the conv-wgrad gather (one entry per output pixel p = (n, ho, wo)):
//   .x = ((n*Hg + ho*sh)*Wg + wo*sw) * ldg      (element offset BEFORE the (-ph, -pw) / tap shift, which is a
//                                                 per-column constant in the GEMM kernel)
//   .y = bit (r*S + s) set  <=>  tap (r, s) of this pixel reads inside the image
// The geometry is fixed per layer, so the table is built once and re-used by every training step; it turns the
// per-step, per-row index arithmetic of the streaming (pixel) dimension into one 8-byte load.
static __global__ void tn_rowtab_kernel(ConvGeom g, int P, int2* __restrict__ tab) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int qw = p % g.Wm;
  const int t = p / g.Wm;
  const int qh = t % g.Hm;
  const int qn = t / g.Hm;
  unsigned mask = 0;
  for (int r = 0; r < g.R; ++r)
    for (int s = 0; s < g.S; ++s) {
      const int hi = qh * g.sh + r * g.dh - g.ph, wi = qw * g.sw + s * g.dw - g.pw;
      if ((unsigned)hi < (unsigned)g.Hg && (unsigned)wi < (unsigned)g.Wg) mask |= 1u << (r * g.S + s);
    }
  tab[p] = make_int2((int)((((long long)qn * g.Hg + qh * g.sh) * g.Wg + qw * g.sw) * g.ldg), (int)mask);
}

template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_t> {
  static constexpr int BP = 64;
  static constexpr int ROW_VECS = 16;   // 16-byte vectors per 128-col row
  static constexpr int ROW_BYTES = 256;
};
template <> struct TnCfg<float> {
  static constexpr int BP = 16;
  static constexpr int ROW_VECS = 32;
  static constexpr int ROW_BYTES = 576;  // 512 + 64 pad: consecutive rows shift by 16 banks
};

__device__ __forceinline__ int tn_hash(int p) { return (p & 3) | (((p >> 3) & 1) << 2); }

template <typename T, int BMODE>
__global__ __launch_bounds__(256) void igemm_tn_kernel(TnArgs a, ConvGeom g) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int BP = TnCfg<T>::BP;
  constexpr int ROW_VECS = TnCfg<T>::ROW_VECS;
  constexpr int ROW_BYTES = TnCfg<T>::ROW_BYTES;
  constexpr int ROWS_PER_PASS = 256 / ROW_VECS;  // 16 (bf16) / 8 (f32)
  constexpr int NI = BP / ROWS_PER_PASS;         // 4 (bf16) / 2 (f32)
  constexpr bool IS_BF16 = (sizeof(T) == 2);

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BP * ROW_BYTES];
  unsigned char* sA = smem;
  unsigned char* sB = smem + BP * ROW_BYTES;

  const int tid = threadIdx.x;
  const int tiles_b = (a.NB + 127) / 128;
  const int tile_b = blockIdx.x % tiles_b, tile_a = blockIdx.x / tiles_b;
  const int na0 = tile_a * 128, nb0 = tile_b * 128;
  const int p_begin = blockIdx.z * a.p_chunk;
  const int p_end = min(a.P, p_begin + a.p_chunk);
  if (p_begin >= p_end) return;

  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;

  const int cc = tid % ROW_VECS;  // 16-byte column chunk
  const int rr = tid / ROW_VECS;
  const int ca = na0 + cc * VEC;  // A column of this thread's vector
  const int cb = nb0 + cc * VEC;  // B column
  const bool ca_ok = ca < a.NA, cb_ok = cb < a.NB;

  int tr = 0, ts = 0, tc = cb;  // conv: tap (r,s) and channel of this thread's B column
  if (BMODE == 1) {
    const int tap = cb / g.Cg;
    tc = cb - tap * g.Cg;
    tr = tap / g.S;
    ts = tap - tr * g.S;
  }

  // LDS byte offset of this thread's vector inside row p
  auto lds_off = [&](int p) -> int {
    if (IS_BF16) {
      const int cp = cc >> 1;
      return p * ROW_BYTES + ((cp ^ tn_hash(p)) << 5) + (cc & 1) * 16;
    } else {
      return p * ROW_BYTES + cc * 16;
    }
  };

  uint4 ra[NI], rb[NI];
  const bool do_colsum = a.colsum != nullptr && tile_b == 0;
  // bias gradient = a column sum over up to 10^5 rows whose terms largely cancel: the f32 parity mode accumulates it
  // in f64 (a sequential f32 chain loses ~sqrt(n) * 6e-8 of SUM|terms|, 0.5 % of the result on the CRNN conv biases)
  typedef typename std::conditional<IS_BF16, float, double>::type CsT;
  CsT csum[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) csum[j] = 0;
  // pixel coordinates of this thread's rows, advanced incrementally by BP per step (no division in the loop)
  int q_n[NI], q_h[NI], q_w[NI];
  if (BMODE == 1) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = p_begin + rr + ROWS_PER_PASS * i;
      q_w[i] = p % g.Wm;
      const int t = p / g.Wm;
      q_h[i] = t % g.Hm;
      q_n[i] = t / g.Hm;
    }
  }
  auto load_tiles = [&](int p0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = p0 + rr + ROWS_PER_PASS * i;
      uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
      if (p < p_end) {
        if (ca_ok) va = ldg16(A + (long long)p * a.lda + ca);
        if (cb_ok) {
          if (BMODE == 0) {
            vb = ldg16(B + (long long)p * a.ldb + cb);
          } else {
            int hi, wi;
            if (conv_src(g, q_h[i], q_w[i], tr, ts, hi, wi))
              vb = ldg16(B + ((long long)(q_n[i] * g.Hg + hi) * g.Wg + wi) * g.ldg + tc);
          }
        }
      }
      ra[i] = va;
      rb[i] = vb;
      if (BMODE == 1) {
        q_w[i] += BP;
        while (q_w[i] >= g.Wm) {
          q_w[i] -= g.Wm;
          if (++q_h[i] == g.Hm) { q_h[i] = 0; ++q_n[i]; }
        }
      }
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wa = wave & 1, wb = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;

  f32x4 acc[4][4];  // [a tile][b tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_tiles(p_begin);
  for (int p0 = p_begin; p0 < p_end; p0 += BP) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int pl = rr + ROWS_PER_PASS * i;
      *(uint4*)(sA + lds_off(pl)) = ra[i];
      *(uint4*)(sB + lds_off(pl)) = rb[i];
      if (do_colsum) {
        const T* pv = (const T*)&ra[i];
#pragma unroll
        for (int j = 0; j < VEC; ++j) csum[j] += to_f32(pv[j]);
      }
    }
    __syncthreads();
    if (p0 + BP < p_end) load_tiles(p0 + BP);

    if constexpr (IS_BF16) {
      // K fragment (8 consecutive p at one column) via two ds_read_b64_tr_b16 per operand tile.
      // Per 16-lane group: lane i supplies the address of row (i>>2), 8-byte piece (i&3) of a
      // [4 rows][16 cols] block; the instruction hands lane i column i of that block.
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
          x[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sA + oa));
          y[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sB + ob));
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

  if (do_colsum) {  // reduce the per-thread column sums over the row groups through LDS
    CsT* red = (CsT*)smem;  // [ROWS_PER_PASS][128]
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[rr * 128 + cc * VEC + j] = csum[j];
    __syncthreads();
    if (tid < 128 && na0 + tid < a.NA) {
      CsT sum = 0;
      for (int r = 0; r < ROWS_PER_PASS; ++r) sum += red[r * 128 + tid];
      int row = na0 + tid;
      if (a.row_perm_h > 0) {
        const int h4 = 4 * a.row_perm_h;
        const int blk = row / h4, rin = row - blk * h4;
        row = blk * h4 + (rin & 3) * a.row_perm_h + (rin >> 2);
      }
      atomicAdd(a.colsum + row, (float)sum);
      if (a.colsum2) atomicAdd(a.colsum2 + row, (float)sum);
    }
  }

  // D[i = a-row][j = b-col]: lane holds rows lg*4+reg, col l15.
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = nb0 + wb * 64 + j * 16 + l15;
      if (col >= a.NB) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int row = na0 + wa * 64 + i * 16 + lg * 4 + q;
        if (row >= a.NA) continue;
        if (a.row_perm_h > 0) {
          const int h4 = 4 * a.row_perm_h;
          const int blk = row / h4, rin = row - blk * h4;
          row = blk * h4 + (rin & 3) * a.row_perm_h + (rin >> 2);
        }
        atomicAdd(a.C + (long long)row * a.ldc + col, acc[i][j][q]);
      }
    }
}

// ---------------------------------------------------------------------------------------------
// TN kernel v2 (bf16): direct-to-LDS staging of both operands, two LDS stages, one barrier per p-step.
// Same LDS image as igemm_tn_kernel ([64 rows(p)][128 cols], 32-byte pieces XOR-swizzled by tn_hash(row)), so
// the ds_read_b64_tr_b16 fragment reads are unchanged; because LDS-DMA writes lane-linearly (one wave
// instruction = 4 full 256-byte rows), the swizzle is applied to the SOURCE column of each lane.
// ---------------------------------------------------------------------------------------------
// BUF: stage through raw buffer resources (glds16_buf) instead of flat pointers + zero page
// ABL (timing only, wrong results): bit 0 = no LDS-DMA in the loop, bit 1 = no fragment reads in the loop,
// bit 2 = no atomic epilogue, bit 3 = no column sums.
// The kernel body: workgroup `bid` of `total` (= tiles * splits) of ONE problem.  igemm_tn_glds_kernel runs it with
// (blockIdx.x, gridDim.x); igemm_tn_glds_grouped_kernel runs SEVERAL problems in one launch (TnGroup), each on its own
// contiguous range of workgroups that starts at a multiple of 8 -- so (bid & 7) is still the XCD of the workgroup.
template <int BMODE, bool BUF, int ABL>
__device__ __forceinline__ void tn_glds_body(const TnArgs& a, const ConvGeom& g, const void* zero, const int bid,
                                             const int total, unsigned char* smem) {
  typedef bf16_t T;
  constexpr int BP = 64, ROW_BYTES = 256, TILE_BYTES = BP * ROW_BYTES;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int tiles_b = (a.NB + 127) / 128, tiles_a = (a.NA + 127) / 128;
  // XCD-aware work map (1-D grid of tiles * splits workgroups).  Hardware workgroup b runs on XCD b % 8.  Virtual
  // order vb = split-major (all tiles of P-chunk 0, then chunk 1, ...); every XCD takes one CONTIGUOUS eighth of
  // it, i.e. whole P-chunks: the workgroups that share an L2 stream through the same rows of dy / x, which are
  // then fetched from HBM about once instead of once per XCD (FETCH_SIZE per launch was ~5x the algorithmic bytes
  // with tiles dealt round-robin over the XCDs).
  const int xq = total >> 3, xr = total & 7, xcd = bid & 7;
  const int vb = xcd * xq + (xcd < xr ? xcd : xr) + (bid >> 3);
  const int ntiles = tiles_a * tiles_b;
  const int split = vb / ntiles, tile = vb - split * ntiles;
  const int tile_b = tile % tiles_b, tile_a = tile / tiles_b;
  const int na0 = tile_a * 128, nb0 = tile_b * 128;
  const int p_begin = split * a.p_chunk;
  const int p_end = min(a.P, p_begin + a.p_chunk);
  if (p_begin >= p_end) return;

  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;
  const int lrow = lane >> 4, pc16 = lane & 15, pp = pc16 >> 1, half = pc16 & 1;

  // This lane stages rows (wave*4+jj)*4 + lrow, jj = 0..3, physical 16-byte chunk pc16.  tn_hash(row) only
  // depends on lrow and bit 1 of (wave*4+jj) = bit 1 of jj, so the lane has two logical columns (h2 = jj>>1).
  //
  // Address generation is the hot part of this kernel (it was VALU-bound: ~400 vector ALU instructions, 70 of them
  // quarter-rate 64-bit multiplies, per 32 MFMAs), so every staged row keeps 32-bit ELEMENT OFFSETS that advance by
  // additions only (operands are < 2^31 elements):
  //   A / dense B: off = p*ld + col                                -> += BP*ld per p-step
  //   conv B:      pix = ((n*Hg + qh*sh)*Wg + qw*sw)*ldg (the window's top-left input pixel, before the tap shift)
  //                -> += BP*sw*ldg per p-step, += wrap_w when the row index wraps, += wrap_h when the image wraps;
  //                the tap (r,s) and channel of this lane's column add a per-column constant tapoff[h2].
  //   conv B with a row table (BMODE 2): pix and the tap-validity mask of every row come from a[].rowtab (one
  //                8-byte load per row, issued one p-step ahead); nothing is computed per step.
  int colA[2], colB[2], tapoff[2], dho[2], dwo[2], tapbit[2];
  bool okA[2], okB[2];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int cp = pp ^ tn_hash((wave * 4 + 2 * h2) * 4 + lrow);
    const int col = (cp * 2 + half) * 8;
    colA[h2] = na0 + col;
    colB[h2] = nb0 + col;
    okA[h2] = colA[h2] < a.NA;
    okB[h2] = colB[h2] < a.NB;
    tapoff[h2] = colB[h2];
    dho[h2] = dwo[h2] = 0;
    tapbit[h2] = 0;
    if (BMODE != 0) {
      const int tap = colB[h2] / g.Cg;
      const int tc = colB[h2] - tap * g.Cg;
      const int tr = tap / g.S, ts = tap - tr * g.S;
      dho[h2] = tr * g.dh - g.ph;  // forward gather (mode 1): hi = q_h*sh + dho, wi = q_w*sw + dwo
      dwo[h2] = ts * g.dw - g.pw;
      tapoff[h2] = (dho[h2] * g.Wg + dwo[h2]) * g.ldg + tc;
      tapbit[h2] = tap;
    }
  }
  int rowoff[4], soffA[4], soffB[4], q_h[4], q_w[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    rowoff[jj] = (wave * 4 + jj) * 4 + lrow;
    const int p = p_begin + rowoff[jj];
    soffA[jj] = (int)((long long)p * a.lda) + colA[jj >> 1];
    q_h[jj] = q_w[jj] = 0;
    if (BMODE == 0) {
      soffB[jj] = (int)((long long)p * a.ldb) + colB[jj >> 1];
    } else if (BMODE == 2) {
      soffB[jj] = 0;
    } else {
      q_w[jj] = p % g.Wm;
      const int t = p / g.Wm;
      q_h[jj] = t % g.Hm;
      const int qn = t / g.Hm;
      soffB[jj] = (int)((((long long)qn * g.Hg + q_h[jj] * g.sh) * g.Wg + q_w[jj] * g.sw) * g.ldg);
    }
  }
  const int stepA = BP * (int)a.lda;
  const int stepB = BMODE == 0 ? BP * (int)a.ldb : BP * g.sw * g.ldg;
  // BMODE 2: row-table entries of the rows this lane stages next (prefetched one p-step ahead)
  int2 ent[4];
  auto fetch_entries = [&](int p0) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int p = p0 + rowoff[jj];
      ent[jj] = p < p_end ? a.rowtab[p] : make_int2(0, 0);
    }
  };
  if (BMODE == 2) fetch_entries(p_begin);
  const int wrap_w = BMODE == 1 ? (g.sh * g.Wg - g.Wm * g.sw) * g.ldg : 0;
  const int wrap_h = BMODE == 1 ? (g.Hg - g.Hm * g.sh) * g.Wg * g.ldg : 0;

  const rsrc_t rsA = make_rsrc(A), rsB = make_rsrc(B);
  auto gload = [&](bool isB, bool ok, int off, void* lds) {
    if (BUF)
      glds16_buf(isB ? rsB : rsA, ok, off, 1, lds);
    else
      glds16(sel_ptr(ok, (isB ? B : A) + off, zero), lds);
  };
  auto stage = [&](unsigned char* sA, int p0) {
    unsigned char* sB = sA + TILE_BYTES;
    const int rows_left = p_end - p0;  // uniform
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int h2 = jj >> 1;
      const bool pv = rowoff[jj] < rows_left;
      gload(false, pv && okA[h2], soffA[jj], sA + (wave * 4 + jj) * 1024);
      soffA[jj] += stepA;
      if (BMODE == 0) {
        gload(true, pv && okB[h2], soffB[jj], sB + (wave * 4 + jj) * 1024);
        soffB[jj] += stepB;
      } else if (BMODE == 2) {
        const bool v = okB[h2] && ((ent[jj].y >> tapbit[h2]) & 1);  // rows beyond p_end carry an empty mask
        gload(true, v, ent[jj].x + tapoff[h2], sB + (wave * 4 + jj) * 1024);
      } else {
        const int hi = q_h[jj] * g.sh + dho[h2], wi = q_w[jj] * g.sw + dwo[h2];
        const bool v = pv && okB[h2] && (unsigned)hi < (unsigned)g.Hg && (unsigned)wi < (unsigned)g.Wg;
        gload(true, v, soffB[jj] + tapoff[h2], sB + (wave * 4 + jj) * 1024);
        q_w[jj] += BP;
        soffB[jj] += stepB;
        while (q_w[jj] >= g.Wm) {
          q_w[jj] -= g.Wm;
          soffB[jj] += wrap_w;
          if (++q_h[jj] == g.Hm) {
            q_h[jj] = 0;
            soffB[jj] += wrap_h;
          }
        }
      }
    }
    if (BMODE == 2) fetch_entries(p0 + BP);
  };

  const int wa = wave & 1, wb = wave >> 1;
  const int l15 = lane & 15, lg = lane >> 4;
  // bias gradient (column sums of the A tile): done by the tile_b == 0 workgroup of every (tile_a, split) row.  Dealing
  // its p-steps round-robin over the tiles_b workgroups (each stages the same A rows) was measured SLOWER (conv wgrad
  // 727 -> 745 us per step: every workgroup then pays the LDS-atomic reduction and 128 global atomics of the epilogue).
  const bool do_colsum = a.colsum != nullptr && tile_b == 0;
  float csum[2][8];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[h2][j] = 0.f;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transpose-read addresses: row p = kk*32 + lg*8 + hh*4 + (l15>>2); tn_hash(p) = (l15>>2) | ((lg&1)<<2) is a lane
  // constant, so each operand tile t needs ONE base offset; kk / hh only add compile-time constants.
  const int hsh = (l15 >> 2) | ((lg & 1) << 2);
  const int rbase = (lg * 8 + (l15 >> 2)) * ROW_BYTES + (l15 & 3) * 8;
  int offA[4], offB[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    offA[t] = rbase + (((wa * 4 + t) ^ hsh) << 5);
    offB[t] = TILE_BYTES + rbase + (((wb * 4 + t) ^ hsh) << 5);
  }

  bf16x8 fa_keep[2][4], fb_keep[2][4];   // ABL bit 1: fragments of the first step, reused
  bool first_compute = true;
  auto compute = [&](const unsigned char* st, int step) {
    if (do_colsum && !(ABL & 8)) {  // bias gradient: column sums of the A tile, re-read from LDS by the lane that staged it
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const uint4 v = *(const uint4*)(st + (wave * 4 + jj) * 1024 + lane * 16);
        const T* pv = (const T*)&v;
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[jj >> 1][j] += to_f32(pv[j]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s16x4 x[2], y[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          x[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(st + offA[t] + kk * 32 * ROW_BYTES + hh * 4 * ROW_BYTES));
          y[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(st + offB[t] + kk * 32 * ROW_BYTES + hh * 4 * ROW_BYTES));
        }
        // concatenate the two 64-bit halves as a vector shuffle (a union round trip made hipcc assemble every
        // fragment with v_mov_b64 pairs: 42 extra VALU per p-step)
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        fa[t] = __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(x[0], x[1], 0, 1, 2, 3, 4, 5, 6, 7));
        fb[t] = __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(y[0], y[1], 0, 1, 2, 3, 4, 5, 6, 7));
        if (ABL & 2) {
          if (first_compute) { fa_keep[kk][t] = fa[t]; fb_keep[kk][t] = fb[t]; }
          else { fa[t] = fa_keep[kk][t]; fb[t] = fb_keep[kk][t]; }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (ABL & 2) first_compute = false;
  };

  const int nsteps = (p_end - p_begin + BP - 1) / BP;
  unsigned char* st0 = smem;
  unsigned char* st1 = smem + 2 * TILE_BYTES;
  stage(st0, p_begin);
  int st = 0;
  for (; st + 1 < nsteps; st += 2) {
    __syncthreads();
    if (!(ABL & 1)) stage(st1, p_begin + (st + 1) * BP);
    compute(st0, st);
    __syncthreads();
    if (!(ABL & 1) && st + 2 < nsteps) stage(st0, p_begin + (st + 2) * BP);
    compute(st1, st + 1);
  }
  if (st < nsteps) {
    __syncthreads();
    compute(st0, st);
  }

  if (do_colsum && !(ABL & 8)) {  // block-level reduction of the per-lane partial column sums with LDS float atomics
    __syncthreads();
    float* red = (float*)smem;
    if (tid < 128) red[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int col = colA[h2] - na0;  // logical column group staged for rows with (jj>>1) == h2
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&red[col + j], csum[h2][j]);
    }
    __syncthreads();
    if (tid < 128 && na0 + tid < a.NA) {
      int row = na0 + tid;
      if (a.row_perm_h > 0) {
        const int h4 = 4 * a.row_perm_h;
        const int blk = row / h4, rin = row - blk * h4;
        row = blk * h4 + (rin & 3) * a.row_perm_h + (rin >> 2);
      }
      atomicAdd(a.colsum + row, red[tid]);
      if (a.colsum2) atomicAdd(a.colsum2 + row, red[tid]);
    }
  }

  if (a.fin == 2) {   // partial tile -> own slab (register order: element (k, tid) = acc tile k of thread tid); tn_finalize_kernel sums
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char*)a.ws + TN_TICKETS * 4, (short)0, 0x7fffffff, 0x00020000);
    const int so = vb * 65536;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, tid * 16, so + (i * 4 + j) * 4096, 0);
    return;
  }
  bool exclusive = false;   // this workgroup holds the tile's complete sum: plain read-modify-write instead of atomics
  if (a.grp > 1 && !(ABL & 4)) {
    const int nsplits = total / ntiles;
    const int g0 = (split / a.grp) * a.grp;
    const int gsize = min(a.grp, nsplits - g0);
    if (gsize > 1) {
      int* tickets = (int*)a.ws;
      const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char*)a.ws + TN_TICKETS * 4, (short)0, 0x7fffffff, 0x00020000);
      const int so = vb * 65536;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, tid * 16, so + (i * 4 + j) * 4096, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      volatile int* bc = (volatile int*)smem;
      int* tk = tickets + (tile * ((nsplits + a.grp - 1) / a.grp) + split / a.grp) % TN_TICKETS;
      if (tid == 0) *bc = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int ticket = *bc;
      if (ticket != gsize - 1) return;
      if (tid == 0) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      for (int jg = 0; jg < gsize; ++jg) {
        const int vbj = (g0 + jg) * ntiles + tile;
        if (vbj == vb) continue;
        const int sj = vbj * 65536;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, sj + (i * 4 + j) * 4096, 16));
      }
      exclusive = gsize == nsplits;
    }
  }

  // epilogue: atomic accumulation (see igemm_tn_kernel)
  if (ABL & 4) {   // keep the accumulators alive without the atomics
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (keep == 123.456f) a.C[0] = keep;
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = nb0 + wb * 64 + j * 16 + l15;
      if (col >= a.NB) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int row = na0 + wa * 64 + i * 16 + lg * 4 + q;
        if (row >= a.NA) continue;
        if (a.row_perm_h > 0) {
          const int h4 = 4 * a.row_perm_h;
          const int blk = row / h4, rin = row - blk * h4;
          row = blk * h4 + (rin & 3) * a.row_perm_h + (rin >> 2);
        }
        float* dst = a.C + (long long)row * a.ldc + col;
        if (exclusive) *dst += acc[i][j][q];
        else atomicAdd(dst, acc[i][j][q]);
      }
    }
}

template <int BMODE, bool BUF = false, int ABL = 0>
__global__ __launch_bounds__(256, 2) void igemm_tn_glds_kernel(TnArgs a, ConvGeom g, const void* zero) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 64 * 256];  // [stage][A|B]
  tn_glds_body<BMODE, BUF, ABL>(a, g, zero, (int)blockIdx.x, (int)gridDim.x, smem);
}

// Grouped launch: up to TN_GROUP_MAX independent weight-gradient problems of one BMODE in ONE launch (mr_tn_defer /
// mr_tn_flush, gemm_conv.hip).  Why: the weight gradients of the small layers (ResNet 1x1 / strided layers at batch 32, the
// LSTM / Linear layers of the CRNN) are launches of 16..64 output tiles whose P loop had to be cut into 8..32 splits to fill
// the chip -- a 2..8-step main loop between an address prologue and a 64 KB atomic epilogue per workgroup.  Several such
// problems side by side fill the chip with 1/8 of the splits each: longer loops, 1/8 of the atomics.  Problem i owns the
// workgroups [blk_end[i-1], blk_end[i]) (each range padded to a multiple of 8: blockIdx & 7 stays the XCD inside a range; the
// pad workgroups leave at once); the struct travels by value in the kernarg segment, indexed with a uniform index.
constexpr int TN_GROUP_MAX = 12;
struct TnGroup {
  TnArgs a[TN_GROUP_MAX];
  ConvGeom g[TN_GROUP_MAX];
  int blk_end[TN_GROUP_MAX];   // exclusive end of problem i's workgroup range (multiple of 8)
  int nblk[TN_GROUP_MAX];      // tiles * splits of problem i (<= the range's length)
  int n;
};
template <int BMODE>
__global__ __launch_bounds__(256, 2) void igemm_tn_glds_grouped_kernel(TnGroup grp, const void* zero) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 64 * 256];
  const int b = (int)blockIdx.x;
  int i = 0;
  while (i + 1 < grp.n && b >= grp.blk_end[i]) ++i;   // uniform: scalar loads from the kernarg segment
  const int b0 = i > 0 ? grp.blk_end[i - 1] : 0;
  const int bid = b - b0;
  if (bid >= grp.nblk[i]) return;
  tn_glds_body<BMODE, true, 0>(grp.a[i], grp.g[i], zero, bid, grp.nblk[i], smem);
}

// C += sum over the splits of a tile's slabs (TnArgs.fin == 2).  One workgroup per (tile, accumulator tile k = i*4 + j); thread tid
// of the slab order = thread tid of the producing workgroup (wave = tid >> 6: wa = wave & 1, wb = wave >> 1), so its f32x4 is
// rows lg*4 .. +3 of column l15 of the 16 x 16 block (i, j) of the wave's 64 x 64 quadrant.
static __global__ __launch_bounds__(256) void tn_finalize_kernel(const f32x4* __restrict__ slabs, float* __restrict__ C, int ntiles,
                                                                int tiles_b, int nsplits, int NA, int NB, int ldc,
                                                                int row_perm_h) {
  const int tile = blockIdx.x >> 4, k = blockIdx.x & 15;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wa = wave & 1, wb = wave >> 1;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < nsplits; ++sp) sum += slabs[((long long)sp * ntiles + tile) * (16 * 256) + k * 256 + tid];
  const int tile_b = tile % tiles_b, tile_a = tile / tiles_b;
  const int i = k >> 2, j = k & 3;
  const int col = tile_b * 128 + wb * 64 + j * 16 + l15;
  if (col >= NB) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int row = tile_a * 128 + wa * 64 + i * 16 + lg * 4 + q;
    if (row >= NA) continue;
    if (row_perm_h > 0) {
      const int h4 = 4 * row_perm_h;
      const int blk = row / h4, rin = row - blk * h4;
      row = blk * h4 + (rin & 3) * row_perm_h + (rin >> 2);
    }
    C[(long long)row * ldc + col] += sum[q];
  }
}

// ---------------------------------------------------------------------------------------------
// TN kernel, big tile: 256 (NA) x 256 (NB) output tile, 8 waves, 64-row p-steps, 128 KB of dynamic LDS (one
// workgroup per CU).  Why: the 128x128 kernel re-reads every operand element from L2 once per tile of the OTHER
// operand -- for the CRNN wgrads that is 2.4 GB of L2->LDS traffic per launch at ~9.5 TB/s, i.e. the kernel is
// bound by L2 bandwidth, not by the MFMA pipe (25 % busy).  A 256x256 tile halves that traffic and gives each
// wave a 128x64 sub-tile (32 MFMAs per 12 transposed fragment reads instead of 16 per 8).
// LDS image: each operand tile is stored as TWO independent 128-column sub-tiles in exactly the layout of
// igemm_tn_glds_kernel ([64 rows][256 B], 32-byte pieces XOR-swizzled by tn_hash(row)); waves 0-3 stage sub-tile 0
// of A and B, waves 4-7 sub-tile 1, with that kernel's lane mapping.  Compute wave (wa, wb): A sub-tile wa (all
// 8 column blocks), B sub-tile wb>>1, column blocks (wb&1)*4 .. +3.
// ---------------------------------------------------------------------------------------------
// SA = number of 128-column A sub-tiles: 2 -> 256x256 tile (wave tile 128x64), 1 -> 128 (NA) x 256 (NB) tile (wave
// tile 64x64; L2 operand traffic 0.75x of the 128x128 kernel).
// Round-2 rewrite: (1) operand staging is the 128x128 kernel's (32-bit element offsets advanced by additions, row table
// for BMODE 2, raw buffer resources: out-of-range -> zeros) -- the first version still carried the 64-bit multiplies
// and zero-page selects; (2) the A fragments are read ONE ROW TILE AHEAD of the MFMAs that consume them: the first
// version read each A fragment inside the row-tile loop right before its 4 MFMAs, so every row tile paid a full LDS
// read latency with nothing to hide it behind (two waves per SIMD) -- measured 3-4x slower than the 128x128 kernel.
// Operands must be < 2 GiB each (buffer-resource num_records); the launcher checks.
template <int BMODE, int SA>
__global__ __launch_bounds__(512) void igemm_tn_big_kernel(TnArgs a, ConvGeom g, const void* zero) {
  typedef bf16_t T;
  constexpr int BP = 64, ROW_BYTES = 256, SUB_BYTES = BP * ROW_BYTES;  // one 128-column sub-tile = 16 KB
  constexpr int STAGE_BYTES = (SA + 2) * SUB_BYTES;                    // [A0 | (A1) | B0 | B1]
  constexpr int TA = 128 * SA, TI = 4 * SA;                            // tile rows (NA), A column blocks per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tn_big[];
  unsigned char* smem = smem_tn_big;
  (void)zero;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int sub = wave >> 2, w4 = wave & 3;  // staging role: sub-tile and wave index inside it
  const int tiles_b = (a.NB + 255) / 256, tiles_a = (a.NA + TA - 1) / TA;
  // XCD-aware work map, see igemm_tn_glds_kernel
  const int total = gridDim.x;
  const int xq = total >> 3, xr = total & 7, xcd = blockIdx.x & 7;
  const int vb = xcd * xq + (xcd < xr ? xcd : xr) + (blockIdx.x >> 3);
  const int ntiles = tiles_a * tiles_b;
  const int split = vb / ntiles, tile = vb - split * ntiles;
  const int tile_b = tile % tiles_b, tile_a = tile / tiles_b;
  const int na0 = tile_a * TA, nb0 = tile_b * 256;
  const int p_begin = split * a.p_chunk;
  const int p_end = min(a.P, p_begin + a.p_chunk);
  if (p_begin >= p_end) return;

  const T* __restrict__ A = (const T*)a.A;
  const T* __restrict__ B = (const T*)a.B;
  const int lrow = lane >> 4, pc16 = lane & 15, pp = pc16 >> 1, half = pc16 & 1;

  // staging state: see igemm_tn_glds_kernel (same lane mapping inside a 128-column sub-tile)
  int colA[2], colB[2], tapoff[2], dho[2], dwo[2], tapbit[2];
  bool okA[2], okB[2];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int cp = pp ^ tn_hash((w4 * 4 + 2 * h2) * 4 + lrow);
    const int col = sub * 128 + (cp * 2 + half) * 8;
    colA[h2] = na0 + col;
    colB[h2] = nb0 + col;
    okA[h2] = colA[h2] < a.NA && sub < SA;
    okB[h2] = colB[h2] < a.NB;
    tapoff[h2] = colB[h2];
    dho[h2] = dwo[h2] = 0;
    tapbit[h2] = 0;
    if (BMODE != 0) {
      const int tap = colB[h2] / g.Cg;
      const int tc = colB[h2] - tap * g.Cg;
      const int tr = tap / g.S, ts = tap - tr * g.S;
      dho[h2] = tr * g.dh - g.ph;
      dwo[h2] = ts * g.dw - g.pw;
      tapoff[h2] = (dho[h2] * g.Wg + dwo[h2]) * g.ldg + tc;
      tapbit[h2] = tap;
    }
  }
  const int ro0 = w4 * 16 + lrow;   // staged rows of this lane: ro0 + 4*jj
  int soffA[4], soffB[4], q_h[4], q_w[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int p = p_begin + ro0 + 4 * jj;
    soffA[jj] = (int)((long long)p * a.lda) + colA[jj >> 1];
    q_h[jj] = q_w[jj] = 0;
    if (BMODE == 0) {
      soffB[jj] = (int)((long long)p * a.ldb) + colB[jj >> 1];
    } else if (BMODE == 2) {
      soffB[jj] = 0;
    } else {
      q_w[jj] = p % g.Wm;
      const int t = p / g.Wm;
      q_h[jj] = t % g.Hm;
      const int qn = t / g.Hm;
      soffB[jj] = (int)((((long long)qn * g.Hg + q_h[jj] * g.sh) * g.Wg + q_w[jj] * g.sw) * g.ldg);
    }
  }
  const int stepA = BP * (int)a.lda;
  const int stepB = BMODE == 0 ? BP * (int)a.ldb : BP * g.sw * g.ldg;
  int2 ent[4];
  auto fetch_entries = [&](int p0) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int p = p0 + ro0 + 4 * jj;
      ent[jj] = p < p_end ? a.rowtab[p] : make_int2(0, 0);
    }
  };
  if (BMODE == 2) fetch_entries(p_begin);
  const int wrap_w = BMODE == 1 ? (g.sh * g.Wg - g.Wm * g.sw) * g.ldg : 0;
  const int wrap_h = BMODE == 1 ? (g.Hg - g.Hm * g.sh) * g.Wg * g.ldg : 0;
  const rsrc_t rsA = make_rsrc(A), rsB = make_rsrc(B);

  auto stage = [&](unsigned char* st, int p0) {
    unsigned char* sA = st + sub * SUB_BYTES;
    unsigned char* sB = st + (SA + sub) * SUB_BYTES;
    const int rows_left = p_end - p0;  // uniform
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int h2 = jj >> 1;
      const bool pv = ro0 + 4 * jj < rows_left;
      if (SA == 2 || sub == 0) {
        glds16_buf(rsA, pv && okA[h2], soffA[jj], 1, sA + (w4 * 4 + jj) * 1024);
        soffA[jj] += stepA;
      }
      if (BMODE == 0) {
        glds16_buf(rsB, pv && okB[h2], soffB[jj], 1, sB + (w4 * 4 + jj) * 1024);
        soffB[jj] += stepB;
      } else if (BMODE == 2) {
        const bool v = okB[h2] && ((ent[jj].y >> tapbit[h2]) & 1);
        glds16_buf(rsB, v, ent[jj].x + tapoff[h2], 1, sB + (w4 * 4 + jj) * 1024);
      } else {
        const int hi = q_h[jj] * g.sh + dho[h2], wi = q_w[jj] * g.sw + dwo[h2];
        const bool v = pv && okB[h2] && (unsigned)hi < (unsigned)g.Hg && (unsigned)wi < (unsigned)g.Wg;
        glds16_buf(rsB, v, soffB[jj] + tapoff[h2], 1, sB + (w4 * 4 + jj) * 1024);
        q_w[jj] += BP;
        soffB[jj] += stepB;
        while (q_w[jj] >= g.Wm) {
          q_w[jj] -= g.Wm;
          soffB[jj] += wrap_w;
          if (++q_h[jj] == g.Hm) {
            q_h[jj] = 0;
            soffB[jj] += wrap_h;
          }
        }
      }
    }
    if (BMODE == 2) fetch_entries(p0 + BP);
  };

  const int wa = wave & 1, wb = wave >> 1;  // compute role: 2 x 4 waves, wave tile (64*SA) (NA) x 64 (NB)
  const int l15 = lane & 15, lg = lane >> 4;
  const bool do_colsum = a.colsum != nullptr && tile_b == 0;
  const bool cs_lane = do_colsum && sub < SA;  // lanes that staged a piece of the A tile
  float* cs_red = (float*)(smem + 2 * STAGE_BYTES);  // [256] column sums of the A tile (bias gradient)
  if (do_colsum && tid < TA) cs_red[tid] = 0.f;       // ordered before its first use by the k-loop's barriers

  f32x4 acc[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses: byte offset of column block `blk` is ((blk ^ hsh) << 5) + rbase.  rbase has bits 5..7 clear, so
  // that equals (rbase | hsh << 5) ^ (blk << 5): ONE lane constant per operand and a v_xor with an immediate per
  // fragment instead of 12 precomputed offsets held in registers across the loop (the 256x256 variant spilled).
  const int hsh = (l15 >> 2) | ((lg & 1) << 2);
  const int rbase = ((lg * 8 + (l15 >> 2)) * ROW_BYTES + (l15 & 3) * 8) | (hsh << 5);
  const int baseA = (SA == 2 ? wa : 0) * SUB_BYTES + rbase;   // sub-tile bases are multiples of 16 KB: XOR-safe
  const int baseB = (SA + (wb >> 1)) * SUB_BYTES + rbase;
  const int cb0 = (wb & 1) * 4;
  const int ca0 = SA == 2 ? 0 : wa * 4;  // first A column block of this wave

  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  auto frag = [&](const unsigned char* stg, int base, int blk, int kk) -> bf16x8 {
    // volatile: hipcc otherwise hoists all 12 XORs out of the p-loop as loop invariants and, at 256 registers,
    // SPILLS them -- each reload then sits behind an s_waitcnt vmcnt(0) that also waits for the LDS-DMA prefetch
    int off;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(off) : "v"(base), "v"(blk << 5));
    const unsigned char* q = stg + off + kk * 32 * ROW_BYTES;
    const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(q));
    const s16x4 x1 =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(q + 4 * ROW_BYTES));
    return __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  auto compute = [&](const unsigned char* st) {
    if (cs_lane) {
      // bias gradient: column sums of the A tile.  Each lane re-reads the 4 chunks it staged, adds the two rows
      // that share a column group, and pushes 16 values per step into the LDS accumulator with float atomics
      // (no live registers across the MFMA section; only the tile_b == 0 workgroups pay for it)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const uint4 v0 = *(const uint4*)(st + sub * SUB_BYTES + (w4 * 4 + 2 * h2) * 1024 + lane * 16);
        const uint4 v1 = *(const uint4*)(st + sub * SUB_BYTES + (w4 * 4 + 2 * h2 + 1) * 1024 + lane * 16);
        const T* p0 = (const T*)&v0;
        const T* p1 = (const T*)&v1;
        const int col = colA[h2] - na0;
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&cs_red[col + j], to_f32(p0[j]) + to_f32(p1[j]));
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fb[t] = frag(st, baseB, cb0 + t, kk);
      bf16x8 fa = frag(st, baseA, ca0, kk);
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        bf16x8 fn = fa;
        if (i + 1 < TI) fn = frag(st, baseA, ca0 + i + 1, kk);   // one row tile ahead of its MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], acc[i][j], 0, 0, 0);
        fa = fn;
      }
    }
  };

  const int nsteps = (p_end - p_begin + BP - 1) / BP;
  unsigned char* st0 = smem;
  unsigned char* st1 = smem + STAGE_BYTES;
  stage(st0, p_begin);
  int st = 0;
  for (; st + 1 < nsteps; st += 2) {
    __syncthreads();
    stage(st1, p_begin + (st + 1) * BP);
    compute(st0);
    __syncthreads();
    if (st + 2 < nsteps) stage(st0, p_begin + (st + 2) * BP);
    compute(st1);
  }
  if (st < nsteps) {
    __syncthreads();
    compute(st0);
  }

  if (do_colsum) {
    __syncthreads();
    float* red = cs_red;
    if (tid < TA && na0 + tid < a.NA) {
      int row = na0 + tid;
      if (a.row_perm_h > 0) {
        const int h4 = 4 * a.row_perm_h;
        const int blk = row / h4, rin = row - blk * h4;
        row = blk * h4 + (rin & 3) * a.row_perm_h + (rin >> 2);
      }
      atomicAdd(a.colsum + row, red[tid]);
      if (a.colsum2) atomicAdd(a.colsum2 + row, red[tid]);
    }
  }

  // epilogue: atomic accumulation (see igemm_tn_kernel)
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = nb0 + wb * 64 + j * 16 + l15;
      if (col >= a.NB) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int row = na0 + wa * (64 * SA) + i * 16 + lg * 4 + q;
        if (row >= a.NA) continue;
        if (a.row_perm_h > 0) {
          const int h4 = 4 * a.row_perm_h;
          const int blk = row / h4, rin = row - blk * h4;
          row = blk * h4 + (rin & 3) * a.row_perm_h + (rin >> 2);
        }
        atomicAdd(a.C + (long long)row * a.ldc + col, acc[i][j][q]);
      }
    }
}

}  // namespace mr
