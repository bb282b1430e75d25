// C-ABI entry points for the MFMA GEMM / implicit-GEMM convolution family.
// Replaces (on MI355X) the cuDNN / cuBLAS calls behind nn.Conv2d, nn.Linear and nn.LSTM's input
// projection on the reference's training path (reference: backbones/crnn.py:44-55,
// backbones/resnet.py:110-256, decoders/crnn.py:8-24).
#include "igemm_core.h"
#include "igemm_p8.h"
#include "tn_taps.h"
#include "nt32.h"
#include "../../include/megreader_hip.h"

#include <algorithm>
#include <mutex>
#include <vector>

namespace mr {

// One 4 KiB page of zeros per device: the source of padded vectors for the direct-to-LDS loads (the only state
// this library keeps).  Created on first use (mr_init() creates it eagerly, e.g. before hipGraph capture).
static const void* zero_page() {
  static std::mutex mu;
  static void* pages[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!pages[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess) return nullptr;
    pages[dev] = p;
  }
  return pages[dev];
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

#define g_nt_variant MR_TUNE(nt_variant)  // 2 = direct-to-LDS kernel (default), 1 = register-staged kernel (A/B debugging)
#define g_nt_deep MR_TUNE(nt_deep)     // 4-stage pipeline of the 4-wave NT kernel: 1 = automatic (low-occupancy launches), 0 never, 2 always

// mr_conv2d_fwd_stats: the f64 column-statistics accumulators the NT launches of the current call attach to their epilogue
// (EpiStore::stats), and whether every launch of the call could (the register-staged fallback kernel cannot)
static thread_local double* g_epi_stats = nullptr;
static thread_local const void* g_epi_addend = nullptr;   // second summand of the NT epilogue for THIS call (mr_conv2d_dgrad_add)
static thread_local bool g_epi_stats_missed = false;
// BatchNorm-backward sums requested for THIS call (mr_conv2d_dgrad_bnb): x / y / mean / rstd of the BatchNorm whose output
// gradient the launch produces; the sums go to g_epi_stats.  Only the 4-wave direct-to-LDS conv kernels (EpiStoreB) serve it.
struct BnbRequest { const void* x; const void* y; const float* mean; const float* rstd; };
static thread_local BnbRequest g_epi_bnb = {nullptr, nullptr, nullptr, nullptr};

// The direct-to-LDS NT kernels address their operands through buffer resources with 2 GiB of records.
template <typename T>
static bool nt_fits_buffer(const NtArgs& a, const ConvGeom& g, int amode) {
  const long long es = sizeof(T);
  const long long bytesA = amode == 0 ? (long long)a.M * a.lda * es
                                      : ((long long)a.M / ((long long)g.Hm * g.Wm) + 1) * g.Hg * g.Wg * g.ldg * es;
  const long long bytesB = (long long)a.N * a.ldb * es;
  return bytesA < (1ll << 31) && bytesB < (1ll << 31);
}

static int num_cus();

// The direct-to-LDS 4-wave kernel for one tile shape and epilogue type: 2-buffer loop, or the 4-buffer loop for low-occupancy
// launches (see the comment inside).
static int num_cus();
// Splits of the reduction for a 4-wave NT launch of `tiles` output tiles and nk_all k-steps (NtArgs.ksplit): only launches that
// keep at most a quarter of the CUs busy on a chain of >= 16 k-steps; as many splits as fill the chip once with >= 4 k-steps
// each, at most 8 (mr_tuning.nt_ksplit: 0 never, 1 automatic, n: at most n).  1 = do not split.
static int nt_split_count(int tiles, int nk_all) {
  const int mode = MR_TUNE(nt_ksplit);
  const int cus = num_cus();
  if (mode <= 0 || tiles * 4 > cus || nk_all < 16 || tiles > NT_SPLIT_TICKETS) return 1;
  int S = cus / tiles;
  if (S > nk_all / 4) S = nk_all / 4;
  if (S > 8) S = 8;
  if (mode > 1 && S > mode) S = mode;
  return S > 1 ? S : 1;
}

template <typename T, int BM, int BN, int AMODE, typename EpiT>
static int launch_nt_glds(const NtArgs& a_in, const ConvGeom& g, const EpiT& epi, int grid, int tiles, hipStream_t stream) {
  constexpr int BK = 8 * VecOf<T>::N;
  NtArgs a2 = a_in;
  unsigned gy = 1;
  if constexpr (sizeof(T) == 2) {
    // Split reduction for launches of a few tiles with a long k-loop (NtArgs.ksplit): at most a quarter of the CUs busy and
    // >= 16 k-steps -> as many splits as keep >= 4 k-steps each, the chip filled once, <= 8.  Needs the per-device
    // split-reduction workspace (mr_set_tn_taps_workspace; launches that use it are stream-ordered with each other).
    const int nk_all = cdiv(a2.K, BK);
    const int S = (a2.m_begin == 0 && (AMODE == 0 || (g.Cg % BK) == 0)) ? nt_split_count(tiles, nk_all) : 1;
    if (S > 1) {
      void* ws = nullptr;
      long long ws_bytes = 0;
      taps_get_workspace(&ws, &ws_bytes);
      const long long need = NT_SPLIT_TICKETS * 4ll + (long long)tiles * S * BM * BN * 4;
      if (ws && need <= ws_bytes && need < (1ll << 31)) {
        a2.ksplit = S;
        a2.ws = ws;
        gy = (unsigned)S;
      }
    }
  }
  if constexpr (sizeof(T) == 2) {
    // low-occupancy launches (about one workgroup per CU or fewer: the small-M layers of the batch-2 detector and the
    // batch-32 recogniser): 4 stage buffers, 3 k-steps of LDS-DMA in flight across the barriers (igemm_nt_glds_kernel NST)
    // Only while every workgroup of the launch is still resident at once with the larger LDS footprint (64 KB for 64x64
    // tiles: two per CU; 96 - 128 KB above: one per CU): 264 tiles of 128x128 on 256 CUs ran 2 % SLOWER with one workgroup
    // per CU and a second round of 8 than as 264 co-resident 2-buffer workgroups (CRNN conv6).  Measured (40 steps, same
    // box): FPN-attention 10.75 -> 10.14 ms, DB 11.61 -> 11.21 ms, Res50-PPM 13.23 -> 13.10 ms.
    const int nk = cdiv(a2.K, BK);
    constexpr int NST = 4;
    constexpr int lds = NST * (BM + BN) * 128;
    const bool deep = g_nt_deep == 2 || (g_nt_deep == 1 && tiles <= num_cus() * ((160 * 1024) / lds) && nk >= 8);
    if (deep && (AMODE == 0 || (g.Cg % BK) == 0)) {
      auto kern = igemm_nt_glds_kernel<T, BM, BN, AMODE, EpiT, NST>;
      static bool attr_set = false;  // per instantiation
      if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
          set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", lds);
          return MR_ERR_LAUNCH;
        }
        attr_set = true;
      }
      hipLaunchKernelGGL(kern, dim3(grid, gy), dim3(256), lds, stream, a2, g, epi);
      MR_CHECK_LAUNCH();
      return MR_OK;
    }
  }
  if (AMODE == 2 && (g.Cg % BK) != 0)
    hipLaunchKernelGGL((igemm_nt_glds_kernel<T, BM, BN, (AMODE == 2 ? 3 : AMODE), EpiT>), dim3(grid), dim3(256), 0, stream,
                       a2, g, epi);
  else
    hipLaunchKernelGGL((igemm_nt_glds_kernel<T, BM, BN, AMODE, EpiT>), dim3(grid, gy), dim3(256), 0, stream, a2, g, epi);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

template <typename T, int BM, int BN, int AMODE>
static int launch_nt_store(const NtArgs& a, const ConvGeom& g, void* C, long long ldc, const float* bias,
                           int relu, hipStream_t stream) {
  EpiStore<T> epi;
  epi.C = (T*)C;
  epi.ldc = ldc;
  epi.bias = bias;
  epi.relu = relu;
  epi.M = a.M;
  epi.N = a.N;
  epi.vec_ok = ((ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0) && ((((uintptr_t)g_epi_addend) & 15) == 0);
  epi.addend = (const T*)g_epi_addend;
  const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
  if constexpr (AMODE != 1) {
    if (g_nt_variant == 2 && nt_fits_buffer<T>(a, g, AMODE)) {
      const int grid = cdiv(cdiv(a.M - a.m_begin, BM), 8) * 8 * cdiv(a.N, BN);  // XCD-aware tile map, see the kernel
      NtArgs a2 = a;
      a2.zero = zero_page();
      if (!a2.zero) { set_error("zero page allocation failed"); return MR_ERR_LAUNCH; }
      epi.stats = g_epi_stats;
      epi.stats_ncopy = MR_BN_COPIES;
      if constexpr (AMODE == 2) {
        if (g_epi_bnb.x != nullptr) {    // statistics epilogue in BatchNorm-backward mode: its own epilogue type
          EpiStoreB<T> eb;
          static_cast<EpiStore<T>&>(eb) = epi;
          eb.bnb_x = (const T*)g_epi_bnb.x;
          eb.bnb_y = (const T*)g_epi_bnb.y;
          eb.bnb_mean = g_epi_bnb.mean;
          eb.bnb_rstd = g_epi_bnb.rstd;
          return launch_nt_glds<T, BM, BN, AMODE, EpiStoreB<T>>(a2, g, eb, grid, tiles, stream);
        }
      }
      return launch_nt_glds<T, BM, BN, AMODE, EpiStore<T>>(a2, g, epi, grid, tiles, stream);
    }
  }
  if (a.m_begin != 0) { set_error("row-range launches need the direct-to-LDS NT kernel"); return MR_ERR_ARG; }
  if (g_epi_stats) g_epi_stats_missed = true;   // this kernel has no statistics epilogue: the caller reduces y itself
  hipLaunchKernelGGL((igemm_nt_kernel<T, BM, BN, AMODE, EpiStore<T>>), dim3(tiles), dim3(256), 0, stream, a, g,
                     epi);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

static int num_cus() {
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

// ---- big-tile (8-wave) NT kernel: bf16, dense or fast-gather conv operands ---------------------------------------
// g_big_mode: 0 = automatic (nt_big_choice), -1 = never, 1 = always 256x256, 2 = always 288x256 (tuning override)
#define g_big_mode MR_TUNE(nt_big)
#ifdef MR_ABLATION
static int g_tn_abl = 0;  // timing-only ablation mask of the TN kernel (mr_set_tn_abl)
#endif
#define g_tn_model MR_TUNE(tn_model)   // 1 (default): the measured split model for the conv wgrad launches too; 0: the old one (A/B)
#define g_tn_splits MR_TUNE(tn_splits)  // > 0: split count override of launch_tn (mr_tuning.tn_splits, tuning only)
#define g_tn_buf MR_TUNE(tn_buf)  // TN kernel staging through buffer resources (mr_tuning.tn_buf); measured 4 % faster
#define g_tn_taps MR_TUNE(tn_taps)  // all-taps wgrad kernel for 3x3 / stride 1 / pad == dilation layers (tn_taps.hip, mr_tuning.tn_taps)
#define g_tn_group MR_TUNE(tn_group)  // GEMM TN kernel split reduction: 0 automatic (slab groups when a workspace is registered), 1 atomics, > 1 forced
#define g_tn_fin MR_TUNE(tn_fin)    // 2: split partials go to slabs with plain stores, a finalize launch sums them (mr_tuning.tn_fin); 0 (default): g_tn_group
#define g_tn_big MR_TUNE(tn_big)  // wide-tile TN kernels: 1 = 256x256, 2 = 128x256 (experimental, see launch_tn), else never

// 0 = use the 4-wave kernels, 1 = 256x256 (8 waves), 3 = 272x256 (8 waves as 1x8); 2 (288x256, spills), 4 / 5 (160x128 with 4 /
// 8 waves) are tuning overrides only.  The big tiles run one
// workgroup per CU, so they only pay when the tile count fills whole rounds of the CUs.  Measured on MI355X (bf16,
// tools/microbench_conv.py --big): 65536x256 (256 tiles, one exact round) K=2304: 653 -> 804 TF/s forward,
// 724 -> 888 TF/s dgrad; 33792x512 is 264 tiles of 256x256 (two rounds, the second 3 % full): 742 -> 604 TF/s, and
// the 288x256 variant that would fit it in one round (236 tiles) spills registers at 12 waves: 742 -> 628 TF/s.
// So: automatic = 256x256 only, and only at >= 85 % CU utilisation.
static int nt_big_choice(int M, int N, int K) {
  if (g_big_mode < 0) return 0;
  if (g_big_mode > 0) return g_big_mode;
  if (N % 256 != 0 || K < MR_TUNE(nt_big_min_k)) return 0;
  const int cus = num_cus();
  const long long tiles = (long long)cdiv(M, 256) * (N / 256);
  const long long rounds = (tiles + cus - 1) / cus;
  const double util = (double)M * N / ((double)rounds * cus * 256 * 256);
  if (util >= 0.85) return 1;
  // 272-row tiles (8 waves as 1x8, 17x2 MFMA tiles per wave): 33792 x 512 is 264 tiles of 256 rows (one round + 8
  // tiles) but 250 tiles of 272 rows -- ONE round, 0.6 % padding.  Measured against the head / tail split
  // (tools/microbench_conv.py --big 3): conv4 fwd 108.8 -> 95.6 us, conv5 fwd 176.6 -> 159.9, conv5 dgrad 167.4 ->
  // 152.8; the tail launch costs ~0.22 of a 256-row round, i.e. ~57 rows of tile height.
  const long long tiles272 = (long long)cdiv(M, 272) * (N / 256);
  const long long rounds272 = (tiles272 + cus - 1) / cus;
  const double util272 = (double)M * N / ((double)rounds272 * cus * 272 * 256);
  if (util272 >= 0.85) {
    const int tiles_n = N / 256;
    double alt = 1e300;   // head / tail cost in rows of tile height, if that split exists
    if (cus % tiles_n == 0) {
      const long long rows_per_round = (long long)(cus / tiles_n) * 256;
      const long long head = (M / rows_per_round) * rows_per_round;
      if (head > 0 && (M - head) * 8 <= M) alt = (double)(head / rows_per_round) * 256 + (head == M ? 0 : 57);
    }
    if ((double)rounds272 * 272 < alt) return 3;
  }
  return 0;
}

// Rows of the head of a head / tail launch (see dispatch_nt_store), 0 = do not split.
static long long nt_head_rows(int M, int N, int K) {
  if (g_big_mode != 0 || N % 256 != 0 || K < MR_TUNE(nt_big_min_k)) return 0;
  const int cus = num_cus();
  const int tiles_n = N / 256;
  if (cus % tiles_n != 0) return 0;
  const long long rows_per_round = (long long)(cus / tiles_n) * 256;
  const long long head = (M / rows_per_round) * rows_per_round;
  return (head > 0 && (M - head) * 8 <= M) ? head : 0;   // tail <= 1/8 of the rows
}

template <typename T, int WM, int WN, int TM, int TN, int AMODE>
static int launch_nt_big(const NtArgs& a, const ConvGeom& g, void* C, long long ldc, const float* bias, int relu,
                         hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  EpiStore<T> epi;
  epi.C = (T*)C;
  epi.ldc = ldc;
  epi.bias = bias;
  epi.relu = relu;
  epi.M = a.M;
  epi.N = a.N;
  epi.vec_ok = ((ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0) && ((((uintptr_t)g_epi_addend) & 15) == 0);
  epi.addend = (const T*)g_epi_addend;
  epi.stats = g_epi_stats;
  epi.stats_ncopy = MR_BN_COPIES;
  NtArgs a2 = a;
  a2.zero = zero_page();
  if (!a2.zero) { set_error("zero page allocation failed"); return MR_ERR_LAUNCH; }
  constexpr size_t lds = 2 * (size_t)(BM + BN) * 128;
  auto kern = igemm_nt_big_kernel<T, WM, WN, TM, TN, AMODE, EpiStore<T>>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", lds);
      return MR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tiles_m = cdiv(a.M - a.m_begin, BM), tiles_n = cdiv(a.N, BN);
  const int grid = cdiv(tiles_m, 8) * 8 * tiles_n;  // XCD-aware map: see the kernel
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, stream, a2, g, epi);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// Phased-schedule 256x256 kernel (igemm_p8.h); g_use_p8: 1 = use it wherever the 8-wave 256x256 kernel would run.
// Measured equal to the v3 kernel (850-900 TF/s on conv3 / conv5, gpurun r2j) -- the schedule is not what bounds the
// tile: its ablation ladder (mr_tuning.nt_p8(2..4), tools/microbench_conv.py --p8) gives 1.16-1.21 PF for the bare
// MFMA + barrier skeleton incl. prologue / epilogue, 1.03 PF with the fragment reads, 0.85-0.90 with the LDS-DMA
// issue on top.  Default: the v3 kernel.
#define g_use_p8 MR_TUNE(nt_p8)
template <typename T, int AMODE>
static int launch_nt_p8(const NtArgs& a, const ConvGeom& g, void* C, long long ldc, const float* bias, int relu,
                        hipStream_t stream) {
  if constexpr (sizeof(T) != 2) {
    return launch_nt_big<T, 2, 4, 8, 4, AMODE>(a, g, C, ldc, bias, relu, stream);
  } else {
    if (!g_use_p8) return launch_nt_big<T, 2, 4, 8, 4, AMODE>(a, g, C, ldc, bias, relu, stream);
    EpiStore<T> epi;
    epi.C = (T*)C;
    epi.ldc = ldc;
    epi.bias = bias;
    epi.relu = relu;
    epi.M = a.M;
    epi.N = a.N;
    epi.vec_ok = ((ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0) && ((((uintptr_t)g_epi_addend) & 15) == 0);
  epi.addend = (const T*)g_epi_addend;
    epi.stats = g_epi_stats;
    epi.stats_ncopy = MR_BN_COPIES;
    NtArgs a2 = a;
    a2.zero = zero_page();
    if (!a2.zero) { set_error("zero page allocation failed"); return MR_ERR_LAUNCH; }
    constexpr size_t lds = 2 * (size_t)(256 + 256) * 128;
    const int tiles_m = cdiv(a.M - a.m_begin, 256), tiles_n = cdiv(a.N, 256);
    const int grid = cdiv(tiles_m, 8) * 8 * tiles_n;
#define MR_P8_LAUNCH(ABL_)                                                                                        \
  {                                                                                                               \
    auto kern = igemm_nt_p8_kernel<AMODE, EpiStore<T>, ABL_>;                                                     \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) {                                                                                              \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=         \
          hipSuccess) {                                                                                           \
        set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", lds);                                      \
        return MR_ERR_LAUNCH;                                                                                     \
      }                                                                                                           \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a2, g, epi);                                     \
  }
#ifdef MR_ABLATION   // measurement-only ablations (mr_tuning.nt_p8(2..4)); wrong results: tools-only build
    if (g_use_p8 == 2) MR_P8_LAUNCH(1)
    else if (g_use_p8 == 3) MR_P8_LAUNCH(2)
    else if (g_use_p8 == 4) MR_P8_LAUNCH(3)
    else
#endif
    MR_P8_LAUNCH(0)
#undef MR_P8_LAUNCH
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
}

// Tile selection shared by every NT launch.  Candidates BM in {128, 96, 64} x BN in {128, 64}; pick the one with
// the smallest modelled time = rounds(tiles / resident slots) * tile work / tile efficiency.  The model exists
// for wave quantisation: e.g. M = 33792 (264 row tiles of 128) x N = 512 gives 1056 tiles on 512 slots = 3 rounds
// with the last one 6 % full, while BM = 96 gives 1408 tiles = 2.75 rounds.
struct TileChoice { int bm, bn; };
// tuning / A-B override (mr_tuning.nt_force_bm / nt_force_bn): one tile shape for every NT launch, bm = 0: the cost model
static inline TileChoice forced_tile() { return TileChoice{MR_TUNE(nt_force_bm), MR_TUNE(nt_force_bn)}; }
static TileChoice nt_tile(int M, int N) {
  if (forced_tile().bm) return forced_tile();
  static const int bms[3] = {128, 96, 64};
  static const int bns[2] = {128, 64};
  // relative MFMA efficiency of a tile shape, calibrated on MI355X with tools/microbench_conv.py --tile
  auto eff = [](int bm, int bn) {
    if (bn == 128) return bm == 128 ? 1.0 : bm == 96 ? 0.90 : 0.85;
    return bm == 128 ? 0.60 : bm == 96 ? 0.60 : 0.55;
  };
  const int cus = num_cus();
  TileChoice best = {128, 128};
  double best_t = 1e300;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 2; ++j) {
      const int bm = bms[i], bn = bns[j];
      if (bn == 128 && N <= 64) continue;
      const long long tiles = (long long)cdiv(M, bm) * cdiv(N, bn);
      const int lds = 2 * (bm + bn) * 128;                  // two k-step buffers
      int per_cu = (160 * 1024) / lds;
      if (per_cu > 3) per_cu = 3;                           // VGPR budget: 3 workgroups of 256 threads
      // greedy makespan: c co-resident workgroups share a CU, each therefore runs c times slower
      long long c = (tiles + cus - 1) / cus;
      if (c > per_cu) c = per_cu;
      const long long rounds = (tiles + cus * c - 1) / (cus * c);
      const double t = (double)rounds * c * bm * bn / eff(bm, bn);
      if (t < best_t) { best_t = t; best.bm = bm; best.bn = bn; }
    }
  return best;
}

// 288x128 tile (8 waves as 2x4, 9x2 MFMA tiles per wave) for problems the 4-wave tiles quantise badly and the 256-column
// big tiles cannot fill: conv4 dgrad of the CRNN (33792 x 256, K = 4608) is 704 tiles of 96x128 = 2 rounds of 512 slots
// with the second 37 % full, but 236 tiles of 288x128 = one round at 92 %: 110.4 -> 92.2 us.  At equal utilisation the
// 4-wave 128x128 kernel (two workgroups per CU) is faster (conv2 dgrad 49.9 vs 55.7 us), hence the quantisation test.
static bool nt_use_288x128(int M, int N, int K) {
  if (g_big_mode != 0 || N % 128 != 0 || K < 2048) return false;
  const int cus = num_cus();
  const long long tiles = (long long)cdiv(M, 288) * (N / 128);
  const long long rounds = (tiles + cus - 1) / cus;
  const double util = (double)M * N / ((double)rounds * cus * 288 * 128);
  if (util < 0.85) return false;
  const TileChoice t = nt_tile(M, N);
  const long long t4 = (long long)cdiv(M, t.bm) * cdiv(N, t.bn);
  int per_cu = (160 * 1024) / (2 * (t.bm + t.bn) * 128);
  if (per_cu > 3) per_cu = 3;
  long long c = (t4 + cus - 1) / cus;
  if (c > per_cu) c = per_cu;
  const long long r4 = (t4 + cus * c - 1) / (cus * c);
  const double quant = (double)t4 / ((double)r4 * cus * c);
  return quant < 0.8;
}

template <typename T, int AMODE>
static int dispatch_nt_store(const NtArgs& a, const ConvGeom& g, void* C, long long ldc, const float* bias,
                             int relu, hipStream_t stream) {
  if constexpr (sizeof(T) == 2 && (AMODE == 0 || AMODE == 2)) {
    constexpr int BK = 8 * VecOf<T>::N;
    // (a launch that also has to produce BatchNorm-backward sums, mr_conv2d_dgrad_bnb, stays on the 4-wave kernels: the
    // 8-wave tiles have no registers to spare for that epilogue)
    if (g_nt_variant == 2 && !forced_tile().bm && (AMODE == 0 || (g.Cg % BK) == 0) && aligned16(C) &&
        nt_fits_buffer<T>(a, g, AMODE) && g_epi_bnb.x == nullptr) {
      const int big = nt_big_choice(a.M, a.N, a.K);
      // Ping-pong 32x32x16 kernel (nt32.hip, round 6; mr_tuning.nt_m32): 1 = wherever the automatic choice takes a 256-column
      // big tile (256x256 -> 256x256, 272x256 -> 288x256), >= 2 = shape (nt_m32 - 1) for every eligible launch (sweeps)
      if (const int m32 = MR_TUNE(nt_m32); m32 != 0 && a.m_begin == 0 && a.ksplit <= 1) {
        const int shape = m32 >= 2 ? m32 - 1 : big == 1 ? 1 : big == 3 ? 2 : 0;
        if (shape) {
          EpiStore<bf16_t> epi;
          epi.C = (bf16_t*)C;
          epi.ldc = ldc;
          epi.bias = bias;
          epi.relu = relu;
          epi.M = a.M;
          epi.N = a.N;
          epi.vec_ok = ((ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0) && ((((uintptr_t)g_epi_addend) & 15) == 0);
          epi.addend = (const bf16_t*)g_epi_addend;
          epi.stats = g_epi_stats;
          epi.stats_ncopy = MR_BN_COPIES;
          return launch_nt32(shape, MR_TUNE(nt_m32_opt), AMODE, a, g, epi, stream);
        }
      }
      if (big == 1) return launch_nt_p8<T, AMODE>(a, g, C, ldc, bias, relu, stream);
      if (big == 2) return launch_nt_big<T, 2, 4, 9, 4, AMODE>(a, g, C, ldc, bias, relu, stream);   // 288x256, 8 waves (2x4)
      if (big == 3) return launch_nt_big<T, 1, 8, 17, 2, AMODE>(a, g, C, ldc, bias, relu, stream);  // 272x256, 8 waves (1x8)
      if (big == 4) return launch_nt_big<T, 2, 2, 5, 4, AMODE>(a, g, C, ldc, bias, relu, stream);   // 160x128, 4 waves (2x2)
      if (big == 5) return launch_nt_big<T, 2, 4, 5, 2, AMODE>(a, g, C, ldc, bias, relu, stream);   // 160x128, 8 waves (2x4)
      if (big == 6) return launch_nt_big<T, 2, 4, 8, 2, AMODE>(a, g, C, ldc, bias, relu, stream);   // 256x128, 8 waves (2x4)
      if (big == 7) return launch_nt_big<T, 2, 4, 9, 2, AMODE>(a, g, C, ldc, bias, relu, stream);   // 288x128, 8 waves (2x4)
      if (big == 8) return launch_nt_big<T, 2, 4, 4, 2, AMODE>(a, g, C, ldc, bias, relu, stream);   // 128x128, 8 waves (2x4)
      if (big == 9) return launch_nt_big<T, 2, 4, 6, 2, AMODE>(a, g, C, ldc, bias, relu, stream);   // 192x128, 8 waves (2x4)
      if (big == 10) return launch_nt_big<T, 4, 2, 2, 2, AMODE>(a, g, C, ldc, bias, relu, stream);  // 128x64, 8 waves (4x2)
      if (big == 11) return launch_nt_big<T, 4, 2, 3, 2, AMODE>(a, g, C, ldc, bias, relu, stream);  // 192x64, 8 waves (4x2)
      if (big == 12) return launch_nt_big<T, 4, 2, 4, 2, AMODE>(a, g, C, ldc, bias, relu, stream);  // 256x64, 8 waves (4x2)
      if (big == 13) return launch_nt_big<T, 2, 4, 3, 2, AMODE>(a, g, C, ldc, bias, relu, stream);  // 96x128, 8 waves (2x4)
      // Head / tail: rows are independent, so a problem whose 256x256 tile count is a few tiles over whole rounds of
      // the CUs (33792 x 512: 264 tiles on 256 CUs) is cut into a head that is EXACTLY whole rounds of big tiles and
      // a tail of the remaining rows for the 4-wave kernel.  No cross-workgroup reduction, two launches.
      const long long head = a.m_begin == 0 ? nt_head_rows(a.M, a.N, a.K) : 0;
      if (head == 0 && a.m_begin == 0 && nt_use_288x128(a.M, a.N, a.K))
        return launch_nt_big<T, 2, 4, 9, 2, AMODE>(a, g, C, ldc, bias, relu, stream);
      if (head > 0) {
        NtArgs ah = a;
        ah.M = (int)head;
        int rc = launch_nt_p8<T, AMODE>(ah, g, C, ldc, bias, relu, stream);
        if (rc != MR_OK || head == a.M) return rc;
        NtArgs at = a;
        at.m_begin = (int)head;
        const TileChoice tt = nt_tile(a.M - (int)head, a.N);
#define MR_NT_TAIL(BM_, BN_) \
  if (tt.bm == BM_ && tt.bn == BN_) return launch_nt_store<T, BM_, BN_, AMODE>(at, g, C, ldc, bias, relu, stream);
        MR_NT_TAIL(128, 128) MR_NT_TAIL(128, 64) MR_NT_TAIL(96, 128) MR_NT_TAIL(96, 64) MR_NT_TAIL(64, 128)
        MR_NT_TAIL(64, 64)
#undef MR_NT_TAIL
      }
      // 8-wave workgroups on the 4-wave tile shapes (mr_tuning.nt_wide8, a bit mask over the shapes): e.g. a 128x128 tile as
      // 2x4 waves of 64x32, two 8-wave workgroups per CU instead of two 4-wave ones -- twice the waves to hide the prologue,
      // the first-operand latency and the epilogue of short-K tiles.  Measured (round 5): CRNN conv1 forward 98.7 -> 80.1 us,
      // dgrad 77.8 -> 72.0 us (tools/microbench_conv.py, profiles/r05_conv_tile_sweep.txt); in the steps, 128x128 + 128x64
      // (mask 3) alone: CRNN 2.658 -> 2.615 ms, Res50-PPM 12.21 -> 11.68, FPN-attention 8.73 -> 8.55, DB 9.85 -> 9.53
      // (profiles/r05_ab_nt_wide8.txt) -- also for the one-round launches the 4-buffer loop used to serve.
      const int w8 = MR_TUNE(nt_wide8);
      // (few-tile launches of these shapes stay on the 8-wave kernels: sending them to the 4-wave kernel's split reduction
      // instead, mr_tuning.nt_ksplit, measured equal or slower -- DB 8.86 vs 8.93 ms, CRNN at 32 crops 1.139 vs 1.148)
      if (w8 > 0 && a.m_begin == 0) {
        const TileChoice tw = nt_tile(a.M, a.N);
#define MR_NT_W8(BIT_, BM_, BN_, WM_, WN_, TM_, TN_) \
  if ((w8 & (BIT_)) && tw.bm == BM_ && tw.bn == BN_) return launch_nt_big<T, WM_, WN_, TM_, TN_, AMODE>(a, g, C, ldc, bias, relu, stream);
        MR_NT_W8(1, 128, 128, 2, 4, 4, 2) MR_NT_W8(2, 128, 64, 4, 2, 2, 2) MR_NT_W8(4, 96, 128, 2, 4, 3, 2)
        MR_NT_W8(8, 64, 128, 2, 4, 2, 2) MR_NT_W8(16, 96, 64, 2, 4, 3, 1) MR_NT_W8(32, 64, 64, 2, 4, 2, 1)
#undef MR_NT_W8
      }
    }
  }
  const TileChoice t = nt_tile(a.M, a.N);
#define MR_NT_CASE(BM_, BN_) \
  if (t.bm == BM_ && t.bn == BN_) return launch_nt_store<T, BM_, BN_, AMODE>(a, g, C, ldc, bias, relu, stream);
  MR_NT_CASE(128, 128) MR_NT_CASE(128, 64) MR_NT_CASE(96, 128) MR_NT_CASE(96, 64) MR_NT_CASE(64, 128)
  MR_NT_CASE(64, 64)
#undef MR_NT_CASE
  set_error("no NT tile for %dx%d", t.bm, t.bn);
  return MR_ERR_ARG;
}

// Wide-tile TN kernel launch: SA = 2 -> 256x256 tile, SA = 1 -> 128 (NA) x 256 (NB) tile; one workgroup per CU.
template <int BMODE, int SA>
static int launch_tn_big(TnArgs a, const ConvGeom& g, int total_steps, hipStream_t stream) {
  const void* z = zero_page();
  if (!z) { set_error("zero page allocation failed"); return MR_ERR_LAUNCH; }
  const int cus = num_cus();
  const int btiles = cdiv(a.NA, 128 * SA) * cdiv(a.NB, 256);
  int bsplits = 1;
  double bbest = 1e300;
  // one workgroup per CU; the epilogue is a full tile of f32 atomics (~8 p-steps of time per 128x128 of it)
  for (int s = 1; s <= 1024 && 2 * s <= total_steps + 1; ++s) {
    const long long blocks = (long long)btiles * s;
    const long long rounds = (blocks + cus - 1) / cus;
    const double cost = (double)rounds * (cdiv(total_steps, s) + 8.0 * SA);
    if (cost < bbest) { bbest = cost; bsplits = s; }
  }
  a.p_chunk = cdiv(cdiv(a.P, bsplits), 64) * 64;
  bsplits = cdiv(a.P, a.p_chunk);
#ifdef MR_ABLATION
  if (g_tn_abl & 8) a.colsum = nullptr;   // timing-only: no fused column sums (wrong bias gradient)
#endif
  constexpr int lds = 2 * (SA + 2) * 64 * 256 + 1024;  // 2 stages x [A.. | B0 | B1] x 16 KB + column-sum accumulator
  auto kern = igemm_tn_big_kernel<BMODE, SA>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", lds);
      return MR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(btiles * bsplits), dim3(512), lds, stream, a, g, z);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// ---- deferred / grouped weight-gradient launches (mr_tn_defer, mr_tn_flush; igemm_tn_glds_grouped_kernel) ----------------
// While deferral is on for the calling host thread, bf16 launches of the 128x128 TN GEMM kernel (dense: BMODE 0, conv with a row
// table: BMODE 2) are RECORDED instead of launched; mr_tn_flush launches everything recorded so far, up to TN_GROUP_MAX problems
// per launch.  The split of every problem is chosen for the GROUP: one p-step count L per workgroup for all problems, so that
// the workgroups of the launch finish together and the chip is filled by problems, not by splits.
#define g_tn_defer MR_TUNE(tn_defer)   // 1 (default): mr_tn_defer(1) records; 0: it is ignored (every launch immediate)
struct TnRecord { TnArgs a; ConvGeom g; };
// Recording is switched per host THREAD (the bracket mr_tn_defer(1) ... mr_tn_defer(0) is one autograd node's), the QUEUE is per
// DEVICE and process-wide behind a mutex: records are pushed on an autograd device worker thread, while the
// end-of-backward callback that flushes them runs on whichever thread completes the graph task -- with a thread-local queue
// that thread saw nothing pending, skipped the launch and still released the operands.
struct TnDeferState {
  bool on = false;
  bool beside = false;          // mr_tn_flush_beside in progress: launches must not touch the shared split-reduction workspace
};
static thread_local TnDeferState g_tn_defer_state;
struct TnQueue { std::vector<TnRecord> q[2]; };   // [0]: BMODE 0, [1]: BMODE 2
static std::mutex g_tn_queue_mutex;
static TnQueue g_tn_queue[MR_MAX_DEVICES];
static TnQueue& tn_queue_of_current_device() {     // (call with g_tn_queue_mutex held)
  int dev = 0;
  (void)hipGetDevice(&dev);
  return g_tn_queue[(unsigned)dev % MR_MAX_DEVICES];
}

template <int BMODE>
static int launch_tn_group(const TnRecord* recs, int n, hipStream_t stream) {
  const void* z = zero_page();
  if (!z) { set_error("zero page allocation failed"); return MR_ERR_LAUNCH; }
  const int cus = num_cus();
  int tiles[TN_GROUP_MAX], steps[TN_GROUP_MAX], max_steps = 1;
  for (int i = 0; i < n; ++i) {
    tiles[i] = cdiv(recs[i].a.NA, 128) * cdiv(recs[i].a.NB, 128);
    steps[i] = cdiv(recs[i].a.P, 64);
    if (steps[i] > max_steps) max_steps = steps[i];
  }
  // L = p-steps per workgroup: makespan model of launch_tn (launch + prologue + atomic epilogue ~ 20 p-steps; two
  // workgroups per CU run their p-steps ~1.3x slower than one)
  int bestL = max_steps;
  double best = 1e300;
  for (int L = 1; L <= max_steps; ++L) {
    long long blocks = 0;
    for (int i = 0; i < n; ++i) blocks += (long long)cdiv(tiles[i] * cdiv(steps[i], L), 8) * 8;
    const long long rounds = (blocks + 2 * cus - 1) / (2 * cus);
    const double cost = (double)rounds * (L + 20.0) * (blocks > cus ? 1.3 : 1.0);
    if (cost < best) { best = cost; bestL = L; }
  }
  TnGroup grp;
  int end = 0;
  for (int i = 0; i < n; ++i) {
    TnArgs a = recs[i].a;
    int splits = cdiv(steps[i], bestL);
    a.p_chunk = cdiv(cdiv(a.P, splits), 64) * 64;
    splits = cdiv(a.P, a.p_chunk);
    a.grp = 1; a.ws = nullptr; a.fin = 0;     // split partials of a grouped launch: f32 atomics (few splits per problem)
    grp.a[i] = a;
    grp.g[i] = recs[i].g;
    grp.nblk[i] = tiles[i] * splits;
    end += cdiv(grp.nblk[i], 8) * 8;
    grp.blk_end[i] = end;
  }
  for (int i = n; i < TN_GROUP_MAX; ++i) { grp.a[i] = grp.a[0]; grp.g[i] = grp.g[0]; grp.nblk[i] = 0; grp.blk_end[i] = end; }
  grp.n = n;
  hipLaunchKernelGGL((igemm_tn_glds_grouped_kernel<BMODE>), dim3(end), dim3(256), 0, stream, grp, z);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

template <typename T, int BMODE>
static int launch_tn_now(TnArgs a, const ConvGeom& g, hipStream_t stream);

static int tn_flush(hipStream_t stream, bool beside = false) {
  TnDeferState& st = g_tn_defer_state;
  const bool was_on = st.on;
  st.on = false;      // single problems below go through launch_tn_now
  st.beside = beside;
  int rc = MR_OK;
  TnQueue mine;       // the device's records, taken out under the lock: launches run without it
  {
    std::lock_guard<std::mutex> lock(g_tn_queue_mutex);
    TnQueue& dq = tn_queue_of_current_device();
    mine.q[0].swap(dq.q[0]);
    mine.q[1].swap(dq.q[1]);
  }
  for (int m = 0; m < 2 && rc == MR_OK; ++m) {
    std::vector<TnRecord>& q = mine.q[m];
    size_t i = 0;
    while (i < q.size() && rc == MR_OK) {
      const int n = (int)std::min<size_t>(TN_GROUP_MAX, q.size() - i);
      if (n == 1)
        rc = m == 0 ? launch_tn_now<bf16_t, 0>(q[i].a, q[i].g, stream) : launch_tn_now<bf16_t, 2>(q[i].a, q[i].g, stream);
      else
        rc = m == 0 ? launch_tn_group<0>(&q[i], n, stream) : launch_tn_group<2>(&q[i], n, stream);
      i += n;
    }
    q.clear();
  }
  st.on = was_on;
  st.beside = false;
  return rc;
}

template <typename T, int BMODE>
static int launch_tn(TnArgs a, const ConvGeom& g, hipStream_t stream) {
  if constexpr (sizeof(T) == 2 && BMODE != 1) {
    TnDeferState& st = g_tn_defer_state;
    if (st.on && g_tn_defer && g_nt_variant == 2 && g_tn_buf && g_tn_big <= 0 && g_tn_fin != 2 && g_tn_splits == 0) {
      const long long bytesA = (long long)a.P * a.lda * 2;
      const long long bytesB = BMODE == 0 ? (long long)a.P * a.ldb * 2
                                          : ((long long)a.P / ((long long)g.Hm * g.Wm) + 1) * g.Hg * g.Wg * g.ldg * 2;
      if (bytesA < (1ll << 31) && bytesB < (1ll << 31)) {
        std::lock_guard<std::mutex> lock(g_tn_queue_mutex);
        tn_queue_of_current_device().q[BMODE == 0 ? 0 : 1].push_back(TnRecord{a, g});
        return MR_OK;
      }
    }
  }
  return launch_tn_now<T, BMODE>(a, g, stream);
}

template <typename T, int BMODE>
static int launch_tn_now(TnArgs a, const ConvGeom& g, hipStream_t stream) {
  constexpr int BP = TnCfg<T>::BP;
  const int tiles = cdiv(a.NA, 128) * cdiv(a.NB, 128);
  // split count: workgroups run 2 per CU and are latency-bound, so time ~ rounds(blocks / 2*CUs) * (p-steps per
  // split + epilogue); the epilogue is a full 128x128 tile of f32 atomics (~8 p-steps of time)
  const int cus = num_cus();
  const int total_steps = cdiv(a.P, BP);
  int splits = 1;
  double best = 1e300;
  for (int s = 1; s <= 1024 && 2 * s <= total_steps + 1; ++s) {
    const long long blocks = (long long)tiles * s;
    const long long rounds = (blocks + 2 * cus - 1) / (2 * cus);
    double cost = (double)rounds * (cdiv(total_steps, s) + 8.0);
    // dense GEMMs (LSTM / Linear weight gradients: 4..64 output tiles) were measured per split count
    // (tools/microbench_tn_dense.py): launch + prologue + atomic epilogue cost ~20 p-steps, not 8, and a workgroup
    // that shares its CU with a second one runs its p-steps ~1.3x slower -- so "one workgroup per CU" wins over
    // "two per CU with half the steps" (dW_hh: 21 us at 16 splits vs 32 us at the 32 the old model chose)
    if (BMODE == 0 || g_tn_model == 1) cost = (double)rounds * (cdiv(total_steps, s) + 20.0) * (blocks > cus ? 1.3 : 1.0);
    if (cost < best) { best = cost; splits = s; }
  }
  if (g_tn_splits > 0) splits = g_tn_splits < total_steps ? g_tn_splits : total_steps;
  if constexpr (sizeof(T) == 2) {
    // wide-tile variants (mode 1: 256x256, mode 2: 128x256; one 8-wave workgroup per CU): they cut the L2 -> LDS
    // operand traffic of the 128x128 kernel to 0.5x / 0.75x.  EXPERIMENTAL, opt-in only (mr_tuning.tn_big): both are
    // bit-for-bit sane (tests/test_kernels_gpu.py) but measured 3-4x SLOWER on MI355X (conv5 wgrad 630 -> 161 /
    // 252 TFLOP/s).  Mode 1 spills ~20 VGPRs inside the k-loop and every scratch reload carries an
    // s_waitcnt vmcnt(0) that also drains the in-flight LDS-DMA prefetch; mode 2 has no spills, so the common
    // cause is elsewhere (one barrier-coupled workgroup per CU with only 32 MFMAs per wave between barriers cannot
    // hide the load latency that two independent 4-wave workgroups hide) -- left for a counter-based look.
    if (g_nt_variant == 2 && g_tn_big > 0) {
      const long long bytesA0 = (long long)a.P * a.lda * 2;
      const long long bytesB0 = BMODE == 0 ? (long long)a.P * a.ldb * 2
                                           : ((long long)a.P / ((long long)g.Hm * g.Wm) + 1) * g.Hg * g.Wg * g.ldg * 2;
      if (bytesA0 < (1ll << 31) && bytesB0 < (1ll << 31)) {   // buffer-resource staging
        const int rc = g_tn_big == 2 ? launch_tn_big<BMODE, 1>(a, g, total_steps, stream)
                                     : launch_tn_big<BMODE, 2>(a, g, total_steps, stream);
        return rc;
      }
    }
  }
  a.p_chunk = cdiv(cdiv(a.P, splits), BP) * BP;
  splits = cdiv(a.P, a.p_chunk);
  if constexpr (sizeof(T) == 2) {
    if (g_nt_variant == 2) {
      // in-launch reduction of the split partials through the all-taps kernel's workspace (TnArgs.grp)
      // Measured (tools/gpu_r2_tn.sh): the publish + ticket round costs ~8 us of latency, the atomics it removes scale with
      // the split count -- conv1 wgrad (5 tiles x ~100 splits) 92.6 -> 80.1 us, but conv6 wgrad (16 splits... of 64 tiles)
      // 44.2 -> 46.8 and the LSTM / Linear weight gradients (4 splits) +-2 us: automatic = only from 16 splits up.
      const bool beside = g_tn_defer_state.beside;   // concurrent with launches of another stream: plain atomics only
      if (g_tn_fin == 2 && splits > 1 && !beside) {   // direct slabs + finalize launch (TnArgs.fin)
        void* ws = nullptr;
        long long ws_bytes = 0;
        taps_get_workspace(&ws, &ws_bytes);
        const long long need = TN_TICKETS * 4ll + (long long)tiles * splits * 65536;
        if (ws && need <= ws_bytes && need < (1ll << 31)) {
          a.fin = 2;
          a.grp = 1;
          a.ws = ws;
        }
      }
      if (a.fin != 2 && g_tn_group != 1 && !beside && splits > 1 && (g_tn_group > 1 || splits >= 16)) {
        void* ws = nullptr;
        long long ws_bytes = 0;
        taps_get_workspace(&ws, &ws_bytes);
        int want = g_tn_group > 1 ? g_tn_group : 4;
        if (want > splits) want = splits;
        const long long need = TN_TICKETS * 4ll + (long long)tiles * splits * 65536;
        if (ws && need <= ws_bytes && need < (1ll << 31) && (long long)tiles * cdiv(splits, want) <= TN_TICKETS) {
          a.grp = want;
          a.ws = ws;
        }
      }
      const void* z = zero_page();
      if (!z) { set_error("zero page allocation failed"); return MR_ERR_LAUNCH; }
      // buffer-resource staging needs every byte offset below the descriptor's 2 GiB num_records
      const long long bytesA = (long long)a.P * a.lda * 2;
      const long long bytesB = BMODE == 0 ? (long long)a.P * a.ldb * 2
                                          : ((long long)a.P / ((long long)g.Hm * g.Wm) + 1) * g.Hg * g.Wg * g.ldg * 2;
#ifdef MR_ABLATION
      if (g_tn_abl && bytesA < (1ll << 31) && bytesB < (1ll << 31)) {   // timing-only ablations (mr_set_tn_abl)
#define MR_TN_ABL(V_) case V_: hipLaunchKernelGGL((igemm_tn_glds_kernel<BMODE, true, V_>), dim3(tiles * splits), \
                                                   dim3(256), 0, stream, a, g, z); break;
        switch (g_tn_abl) { MR_TN_ABL(1) MR_TN_ABL(2) MR_TN_ABL(3) MR_TN_ABL(4) MR_TN_ABL(8) MR_TN_ABL(15) default: break; }
#undef MR_TN_ABL
      } else
#endif
      if (g_tn_buf && bytesA < (1ll << 31) && bytesB < (1ll << 31))
        hipLaunchKernelGGL((igemm_tn_glds_kernel<BMODE, true>), dim3(tiles * splits), dim3(256), 0, stream, a, g, z);
      else
        hipLaunchKernelGGL((igemm_tn_glds_kernel<BMODE, false>), dim3(tiles * splits), dim3(256), 0, stream, a, g, z);
      MR_CHECK_LAUNCH();
      if (a.fin == 2) {
        hipLaunchKernelGGL(tn_finalize_kernel, dim3(tiles * 16), dim3(256), 0, stream,
                           (const f32x4*)((const char*)a.ws + TN_TICKETS * 4), a.C, tiles, cdiv(a.NB, 128), splits, a.NA, a.NB,
                           a.ldc, a.row_perm_h);
        MR_CHECK_LAUNCH();
      }
      return MR_OK;
    }
  }
  hipLaunchKernelGGL((igemm_tn_kernel<T, BMODE>), dim3(tiles, 1, splits), dim3(256), 0, stream, a, g);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // namespace mr

using namespace mr;

namespace mr {
bool gemm_nt_skinny(int dtype, const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc,
                    const float* bias, int relu, int M, int N, int K, hipStream_t stream);
}

// ---- conv + bias + ReLU + max-pool in one launch (EpiPool, igemm_core.h; round 6, VERDICT r5 task 8) -----------------------
// Which 8-wave tile serves the fused launch: the one the plain forward would take, if its rows can be cut on window-row / image
// boundaries.  Returns 0 (not eligible), 1 = 256x256, 3 = 272x256 (1x8 waves), 8 = 128x128 (two 8-wave workgroups per CU) and the
// rows of M a tile covers in *tile_rows.
static int conv_pool_plan(int M, int N, int K, int Cin, int Ho, int Wo, int kh, int sh, int ph, int* tile_rows) {
  if (kh != sh || ph != 0 || Ho % kh != 0 || (Cin % 64) != 0 || g_nt_variant != 2) return 0;
  const int img = Ho * Wo, wrow = kh * Wo;
  auto fit = [&](int bm) {      // largest multiple of a window row that fills at most bm rows and tiles the images exactly
    int t = (bm / wrow) * wrow;
    if (t <= 0) return 0;
    if (t >= img) t = (t / img) * img;
    else while (t > 0 && img % t != 0) t -= wrow;
    return t;
  };
  const int big = nt_big_choice(M, N, K);
  if (big == 1 || big == 3) {
    const int bm = big == 1 ? 256 : 272;
    const int t = fit(bm);
    if (t > 0 && N % 256 == 0 && t * 16 >= bm * 15) { *tile_rows = t; return big; }   // (at most 1/16 of a tile's rows idle)
  }
  const TileChoice tw = nt_tile(M, N);
  if (tw.bm == 128 && tw.bn == 128 && (MR_TUNE(nt_wide8) & 1) && N % 128 == 0) {
    const int t = fit(128);
    if (t == 128) { *tile_rows = t; return 8; }
  }
  return 0;
}

template <int WM, int WN, int TM, int TN>
static int launch_nt_pool(const NtArgs& a, const ConvGeom& g, const EpiPool<bf16_t>& epi, hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr size_t lds_stage = 2 * (size_t)(BM + BN) * 128, lds_tile = (size_t)BM * BN * 2;
  constexpr size_t lds = lds_stage > lds_tile ? lds_stage : lds_tile;
  auto kern = igemm_nt_big_kernel<bf16_t, WM, WN, TM, TN, 2, EpiPool<bf16_t>>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", lds);
      return MR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tiles_m = cdiv(a.M, a.tile_rows), tiles_n = cdiv(a.N, BN);
  const int grid = cdiv(tiles_m, 8) * 8 * tiles_n;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, stream, a, g, epi);
  MR_CHECK_LAUNCH();
  return MR_OK;
}


extern "C" {

// Eagerly create per-device state (the zero page).  Call once per device before capturing a hipGraph.
int mr_init(void) {
  const int rc = tuning_from_env();   // MEGREADER_TUNING="field=value,..." (include/megreader_hip.h: mr_tuning), applied once
  if (rc != MR_OK) return rc;
  if (!zero_page()) { set_error("mr_init: zero page allocation failed"); return MR_ERR_LAUNCH; }
  return MR_OK;
}

// The A/B and tuning switches of the kernels in this file live in mr_tuning (tuning.hip, include/megreader_hip.h); only the
// timing-only ablation masks of the -DMR_ABLATION tools build keep setters of their own (include/megreader_hip_ablation.h).
#ifdef MR_ABLATION
int mr_set_tn_abl(int mask) {
  const int old = g_tn_abl;
  g_tn_abl = mask;
  return old;
}
int mr_set_tn_taps_abl(int mask) { return taps_set_abl(mask); }
#endif

// Workspace of the all-taps kernel's in-launch split reduction: `bytes` of device memory (16 KB of tickets + 147456 B
// per workgroup of the largest launch, i.e. 2 * CUs slabs), ZEROED by the caller once; NULL / 0 withdraws it (the
// kernel then reduces with f32 atomics only).  One workspace per device (the current device at the time of the call).
// Launches that use it must be stream-ordered with each other.
int mr_set_tn_taps_workspace(void* ws, long long bytes) {
  MR_CHECK_ARG((ws == nullptr) == (bytes == 0) && bytes >= 0 && (((uintptr_t)ws) & 15) == 0,
               "mr_set_tn_taps_workspace: bad workspace");
  taps_set_workspace(ws, bytes);
  return MR_OK;
}

// host only: 1 when mr_conv2d_wgrad_tab would run the all-taps kernel for this geometry (bf16, row table of
// N*Ho*Wo*8 bytes) under the current mr_tuning.tn_taps setting, else 0
int mr_tn_taps_would_run(int Nimg, int H, int W, int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw,
                         int ph, int pw, int dh, int dw, int Ho, int Wo) {
  return (g_tn_taps && g_nt_variant == 2 &&
          taps_eligible(Nimg, H, W, Cin, ldx, Cout, lddy, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo,
                        (long long)Nimg * Ho * Wo * 8)) ? 1 : 0;
}

// Which NT tile (BM*1000 + BN) a problem of M rows x N columns is dispatched to (profiling / bench labels).
int mr_nt_tile_code(int M, int N) {
  const TileChoice t = nt_tile(M, N);
  return t.bm * 1000 + t.bn;
}

// Same query including the big-tile policy: cg = channel count of the gathered conv operand (0 for a dense GEMM).
// Returns 256256 when the 8-wave 256x256 kernel would run for a bf16 problem of this shape.
int mr_nt_kernel_code(int dtype, int M, int N, int K, int cg) {
  if (dtype == MR_BF16 && g_nt_variant == 2 && !forced_tile().bm && (cg == 0 || cg % 64 == 0)) {
    const int big = nt_big_choice(M, N, K);
    if (big == 1) return 256256;
    if (big == 2) return 288256;
    if (big == 3) return 272256;
    if (big == 4 || big == 5) return 160128;
    if (big == 6) return 256128;
    if (big == 7) return 288128;
    if (nt_head_rows(M, N, K) > 0) return 256257;   // TWO launches: head (whole rounds of 256x256 tiles) + a 4-wave tail
    if (nt_use_288x128(M, N, K)) return 288128;
  }
  return mr_nt_tile_code(M, N);
}

int mr_gemm_nt(int dtype, const void* A, long long lda, const void* B, int ldb, void* C, long long ldc,
               const float* bias, int relu, int M, int N, int K, hipStream_t stream) {
  MR_CHECK_ARG(dtype == MR_F32 || dtype == MR_BF16, "mr_gemm_nt: bad dtype %d", dtype);
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(M > 0 && N > 0 && K >= 0, "mr_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  MR_CHECK_ARG(K % vec == 0 && lda % vec == 0 && ldb % vec == 0,
               "mr_gemm_nt: K/lda/ldb must be multiples of %d (K=%d lda=%lld ldb=%d)", vec, K, lda, ldb);
  MR_CHECK_ARG(aligned16(A) && aligned16(B), "mr_gemm_nt: A and B must be 16-byte aligned");
  if (mr::gemm_nt_skinny(dtype, A, lda, B, ldb, C, ldc, bias, relu, M, N, K, stream)) {   // M <= 32: gemm_skinny.hip
    MR_CHECK_LAUNCH();
    return MR_OK;
  }
  NtArgs a;
  a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.zero = nullptr;
  ConvGeom g = {};
  if (dtype == MR_F32) return dispatch_nt_store<float, 0>(a, g, C, ldc, bias, relu, stream);
  return dispatch_nt_store<bf16_t, 0>(a, g, C, ldc, bias, relu, stream);
}

// Deferred weight-gradient launches.  mr_tn_defer(1): from now on, on the calling host thread, the bf16 weight-gradient GEMMs that
// would run on the 128x128 TN kernel (mr_gemm_tn; mr_conv2d_wgrad_tab off the all-taps kernel) are recorded instead of launched;
// mr_tn_defer(0) stops recording (what is recorded stays).  mr_tn_flush(stream) launches everything recorded, several problems
// per launch.  The CALLER keeps every operand alive and unmodified until the flush and must not read an output before it.
// Returns the previous setting.  Host only.
int mr_tn_defer(int on) {
  const int old = g_tn_defer_state.on ? 1 : 0;
  g_tn_defer_state.on = on != 0;
  return old;
}
int mr_tn_pending(void) {
  std::lock_guard<std::mutex> lock(g_tn_queue_mutex);
  const TnQueue& dq = tn_queue_of_current_device();
  return (int)(dq.q[0].size() + dq.q[1].size());
}
// Drops the current device's recorded problems WITHOUT launching them (a backward pass that raised: its records point at operands
// the caller is about to release) and switches recording off for the calling thread; returns how many were dropped.  Host only.
int mr_tn_discard(void) {
  g_tn_defer_state.on = false;
  std::lock_guard<std::mutex> lock(g_tn_queue_mutex);
  TnQueue& dq = tn_queue_of_current_device();
  const int n = (int)(dq.q[0].size() + dq.q[1].size());
  dq.q[0].clear();
  dq.q[1].clear();
  return n;
}
int mr_tn_flush(hipStream_t stream) { return tn_flush(stream); }
// The same on a stream that runs BESIDE the stream of the other weight-gradient launches: single leftover problems reduce their
// split partials with plain f32 atomics (the ticket / slab workspace of the in-launch reduction is shared per device and
// must only be used by launches that are stream-ordered with each other).
int mr_tn_flush_beside(hipStream_t stream) { return tn_flush(stream, true); }

int mr_gemm_tn(int dtype, const void* A, long long lda, const void* B, long long ldb, float* C, int ldc, int P,
               int NA, int NB, int row_perm_h, float* colsum, hipStream_t stream) {
  return mr_gemm_tn2(dtype, A, lda, B, ldb, C, ldc, P, NA, NB, row_perm_h, colsum, nullptr, stream);
}

// mr_gemm_tn with a second destination for the column sums (both receive += sum_p A[p, :]): the two bias vectors of an LSTM
// direction (b_ih, b_hh: nn.LSTM keeps two, their gradients are equal) take them straight from the weight-gradient launch.
int mr_gemm_tn2(int dtype, const void* A, long long lda, const void* B, long long ldb, float* C, int ldc, int P,
                int NA, int NB, int row_perm_h, float* colsum, float* colsum2, hipStream_t stream) {
  MR_CHECK_ARG(dtype == MR_F32 || dtype == MR_BF16, "mr_gemm_tn: bad dtype %d", dtype);
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(P > 0 && NA > 0 && NB > 0, "mr_gemm_tn: bad shape P=%d NA=%d NB=%d", P, NA, NB);
  // NA need not be a vector multiple as long as the rows of A are (lda >= NA rounded up): the last 16-byte chunk of a row
  // then reads A's padding columns, whose products land in output rows >= NA, which are never stored (a Linear layer
  // with 38 outputs keeps its gradient rows in a 40-column buffer and accumulates straight into the [38, K] sink)
  MR_CHECK_ARG(NB % vec == 0 && lda % vec == 0 && ldb % vec == 0 && lda >= (NA + vec - 1) / vec * vec,
               "mr_gemm_tn: NB/lda/ldb must be multiples of %d and lda >= NA rounded up to it", vec);
  MR_CHECK_ARG(aligned16(A) && aligned16(B), "mr_gemm_tn: A and B must be 16-byte aligned");
  MR_CHECK_ARG(row_perm_h == 0 || NA % (4 * row_perm_h) == 0, "mr_gemm_tn: NA must be a multiple of 4*row_perm_h");
  TnArgs a;
  a.A = A; a.B = B; a.C = C; a.P = P; a.NA = NA; a.NB = NB; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.p_chunk = 0; a.row_perm_h = row_perm_h; a.colsum = colsum; a.rowtab = nullptr; a.grp = 1; a.ws = nullptr;
  a.colsum2 = colsum ? colsum2 : nullptr;
  ConvGeom g = {};
  if (dtype == MR_F32) return launch_tn<float, 0>(a, g, stream);
  return launch_tn<bf16_t, 0>(a, g, stream);
}

static int fill_geom(ConvGeom& g, int mode, int Hg, int Wg, int Cg, int ldg, int Hm, int Wm, int R, int S, int sh,
                     int sw, int ph, int pw, int dh, int dw) {
  g.Hg = Hg; g.Wg = Wg; g.Cg = Cg; g.ldg = ldg; g.Hm = Hm; g.Wm = Wm; g.R = R; g.S = S;
  g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw; g.dh = dh; g.dw = dw; g.mode = mode;
  return 0;
}

// y[n,ho,wo,k] = act(bias[k] + sum_{r,s,c} x[n, ho*sh-ph+r*dh, wo*sw-pw+s*dw, c] * w[k,r,s,c])
int mr_conv2d_fwd(int dtype, const void* x, const void* w_krsc, const float* bias, void* y, int relu, int Nimg,
                  int H, int W, int Cin, int ldx, int Cout, int ldy, int R, int S, int sh, int sw, int ph, int pw,
                  int dh, int dw, int Ho, int Wo, hipStream_t stream) {
  MR_CHECK_ARG(dtype == MR_F32 || dtype == MR_BF16, "mr_conv2d_fwd: bad dtype %d", dtype);
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(Cin % vec == 0 && ldx % vec == 0, "mr_conv2d_fwd: Cin (%d) and ldx (%d) must be multiples of %d",
               Cin, ldx, vec);
  MR_CHECK_ARG(aligned16(x) && aligned16(w_krsc), "mr_conv2d_fwd: x and w must be 16-byte aligned");
  MR_CHECK_ARG((long long)Nimg * Ho * Wo < (1ll << 31), "mr_conv2d_fwd: too many output pixels");
  MR_CHECK_ARG(Ho == (H + 2 * ph - dh * (R - 1) - 1) / sh + 1 && Wo == (W + 2 * pw - dw * (S - 1) - 1) / sw + 1,
               "mr_conv2d_fwd: output size %dx%d inconsistent with geometry", Ho, Wo);
  NtArgs a;
  a.A = x; a.B = w_krsc; a.M = Nimg * Ho * Wo; a.N = Cout; a.K = R * S * Cin; a.lda = 0; a.ldb = R * S * Cin; a.zero = nullptr;
  ConvGeom g;
  fill_geom(g, 1, H, W, Cin, ldx, Ho, Wo, R, S, sh, sw, ph, pw, dh, dw);
  if (R * S <= 32) {
    if (dtype == MR_F32) return dispatch_nt_store<float, 2>(a, g, y, ldy, bias, relu, stream);
    return dispatch_nt_store<bf16_t, 2>(a, g, y, ldy, bias, relu, stream);
  }
  if (dtype == MR_F32) return dispatch_nt_store<float, 1>(a, g, y, ldy, bias, relu, stream);
  return dispatch_nt_store<bf16_t, 1>(a, g, y, ldy, bias, relu, stream);
}

// host only: 1 when mr_conv2d_fwd_pool serves this geometry (else run mr_conv2d_fwd + mr_maxpool_fwd)
int mr_conv2d_fwd_pool_ok(int dtype, int Nimg, int H, int W, int Cin, int ldx, int Cout, int R, int S, int sh, int sw, int ph, int pw,
                          int dh, int dw, int Ho, int Wo, int pkh, int pkw, int psh, int psw, int pph, int ppw) {
  (void)H; (void)W; (void)ppw; (void)pkw; (void)psw;
  if (dtype != MR_BF16 || R * S > 32 || Cin != ldx || Cout % 8 != 0 || sh != 1 || sw != 1 || dh != 1 || dw != 1 || ph < 0 || pw < 0)
    return 0;
  int tr = 0;
  return conv_pool_plan(Nimg * Ho * Wo, Cout, R * S * Cin, Cin, Ho, Wo, pkh, psh, pph, &tr) ? 1 : 0;
}

// y_pool [Nimg, PHo, PWo, Cout] (bf16) = maxpool_{pkh x pkw, stride (psh, psw), pad (pph, ppw)}(relu?(conv(x) + bias)), idx = the
// arg-max codes mr_maxpool_fwd would store for it (one byte per element).  Replaces the pair mr_conv2d_fwd + mr_maxpool_fwd for
// the geometries mr_conv2d_fwd_pool_ok accepts: the full-resolution activation is neither written nor re-read.
int mr_conv2d_fwd_pool(int dtype, const void* x, const void* w_krsc, const float* bias, void* y_pool, unsigned char* idx, int relu,
                       int Nimg, int H, int W, int Cin, int ldx, int Cout, int R, int S, int sh, int sw, int ph, int pw, int dh,
                       int dw, int Ho, int Wo, int pkh, int pkw, int psh, int psw, int pph, int ppw, int PHo, int PWo,
                       hipStream_t stream) {
  MR_CHECK_ARG(mr_conv2d_fwd_pool_ok(dtype, Nimg, H, W, Cin, ldx, Cout, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo, pkh, pkw, psh, psw, pph,
                                     ppw),
               "mr_conv2d_fwd_pool: geometry not served (ask mr_conv2d_fwd_pool_ok)");
  MR_CHECK_ARG(PHo == Ho / pkh && PWo == (Wo + 2 * ppw - pkw) / psw + 1 && pkh * pkw <= 255,
               "mr_conv2d_fwd_pool: pooled size %dx%d inconsistent", PHo, PWo);
  MR_CHECK_ARG(aligned16(x) && aligned16(w_krsc) && aligned16(y_pool) && ((uintptr_t)idx & 7) == 0, "mr_conv2d_fwd_pool: alignment");
  NtArgs a;
  a.A = x; a.B = w_krsc; a.M = Nimg * Ho * Wo; a.N = Cout; a.K = R * S * Cin; a.lda = 0; a.ldb = R * S * Cin; a.zero = zero_page();
  ConvGeom g;
  fill_geom(g, 1, H, W, Cin, ldx, Ho, Wo, R, S, sh, sw, ph, pw, dh, dw);
  MR_CHECK_ARG(nt_fits_buffer<bf16_t>(a, g, 2), "mr_conv2d_fwd_pool: operands must be smaller than 2 GiB");
  const int plan = conv_pool_plan(a.M, a.N, a.K, Cin, Ho, Wo, pkh, psh, pph, &a.tile_rows);
  EpiPool<bf16_t> epi;
  epi.C = (bf16_t*)y_pool; epi.idx = idx; epi.ldc = Cout; epi.bias = bias; epi.relu = relu; epi.M = a.M; epi.N = Cout;
  epi.Nimg = Nimg; epi.Ho = Ho; epi.Wo = Wo; epi.kh = pkh; epi.kw = pkw; epi.sw = psw; epi.pw = ppw; epi.PHo = PHo; epi.PWo = PWo;
  if (plan == 1) return launch_nt_pool<2, 4, 8, 4>(a, g, epi, stream);
  if (plan == 3) return launch_nt_pool<1, 8, 17, 2>(a, g, epi, stream);
  return launch_nt_pool<2, 4, 4, 2>(a, g, epi, stream);
}

// mr_conv2d_fwd + the BatchNorm batch statistics of its output: bn_sums (f64 [MR_BN_COPIES][2][Cout], zeroed by the caller;
// mr_bn_scratch_doubles(Cout) doubles as handed to mr_bn_fwd_train) receives sum_p y[p][c] and sum_p y[p][c]^2 over the
// values as stored.  The direct-to-LDS NT kernels accumulate them in their epilogue (no separate pass over y); if a launch
// had to use the register-staged kernel, one reduction pass over y runs here instead -- the contract is the same either way.
int mr_conv2d_fwd_stats(int dtype, const void* x, const void* w_krsc, const float* bias, void* y, double* bn_sums,
                        int Nimg, int H, int W, int Cin, int ldx, int Cout, int R, int S, int sh, int sw, int ph, int pw,
                        int dh, int dw, int Ho, int Wo, hipStream_t stream) {
  MR_CHECK_ARG(bn_sums != nullptr, "mr_conv2d_fwd_stats: bn_sums is null");
  g_epi_stats = bn_sums;
  g_epi_stats_missed = false;
  const int rc = mr_conv2d_fwd(dtype, x, w_krsc, bias, y, 0, Nimg, H, W, Cin, ldx, Cout, Cout, R, S, sh, sw, ph, pw, dh, dw,
                               Ho, Wo, stream);
  const bool missed = g_epi_stats_missed;
  g_epi_stats = nullptr;
  g_epi_stats_missed = false;
  if (rc != MR_OK) return rc;
  if (missed) return mr_bn_stats(dtype, y, bn_sums, (long long)Nimg * Ho * Wo, Cout, stream);
  return MR_OK;
}

// dx[n,h,w,c] = sum_{r,s,k} dy[n,(h+ph-r*dh)/sh,(w+pw-s*dw)/sw,k] * w[k,r,s,c]; w_crsk = w transposed to [c][r][s][k]
int mr_conv2d_dgrad(int dtype, const void* dy, const void* w_crsk, void* dx, int Nimg, int H, int W, int Cin,
                    int lddx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw,
                    int Ho, int Wo, hipStream_t stream) {
  MR_CHECK_ARG(dtype == MR_F32 || dtype == MR_BF16, "mr_conv2d_dgrad: bad dtype %d", dtype);
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(Cout % vec == 0 && lddy % vec == 0, "mr_conv2d_dgrad: Cout/lddy must be multiples of %d", vec);
  MR_CHECK_ARG(aligned16(dy) && aligned16(w_crsk), "mr_conv2d_dgrad: dy and w must be 16-byte aligned");
  NtArgs a;
  a.A = dy; a.B = w_crsk; a.M = Nimg * H * W; a.N = Cin; a.K = R * S * Cout; a.lda = 0; a.ldb = R * S * Cout; a.zero = nullptr;
  ConvGeom g;
  fill_geom(g, 2, Ho, Wo, Cout, lddy, H, W, R, S, sh, sw, ph, pw, dh, dw);
  if (R * S <= 32 && sh == 1 && sw == 1) {
    if (dtype == MR_F32) return dispatch_nt_store<float, 2>(a, g, dx, lddx, nullptr, 0, stream);
    return dispatch_nt_store<bf16_t, 2>(a, g, dx, lddx, nullptr, 0, stream);
  }
  if (dtype == MR_F32) return dispatch_nt_store<float, 1>(a, g, dx, lddx, nullptr, 0, stream);
  return dispatch_nt_store<bf16_t, 1>(a, g, dx, lddx, nullptr, 0, stream);
}

// mr_conv2d_dgrad with a second summand in the epilogue: dx = dgrad(dy, w) + addend (addend: NHWC like dx, same lddx).  A
// ResNet block's input receives its gradient from two branches (reference backbones/resnet.py:152-181: `out += residual`): the
// residual branch's gradient rides in the epilogue of the dgrad of the block's first convolution instead of an elementwise add
// kernel over the block input (18 per ResNet-50 step).  addend may alias dx (each element is read before it is written by the
// same lane); null = plain mr_conv2d_dgrad.
int mr_conv2d_dgrad_add(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, int Nimg, int H, int W,
                        int Cin, int lddx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw,
                        int Ho, int Wo, hipStream_t stream) {
  g_epi_addend = addend;
  const int rc = mr_conv2d_dgrad(dtype, dy, w_crsk, dx, Nimg, H, W, Cin, lddx, Cout, lddy, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo,
                                 stream);
  g_epi_addend = nullptr;
  return rc;
}

// mr_conv2d_dgrad_add that ALSO leaves the BatchNorm-backward sums of the gradient it produces: dx is the gradient of the output
// y = act(bn(x) [+ residual]) of a training-mode BatchNorm (reference nn.Sequential(conv, bn, relu) chains, backbones/resnet.py:
// 113-181), and the BatchNorm's backward needs  sum g'  and  sum g' xhat  per channel (g' = dx where y > 0).  They are accumulated
// in the dgrad's epilogue into bn_sums (layout and zeroing as for mr_bn_bwd's scratch: f64 [MR_BN_COPIES][2][Cin], zeroed by the
// caller), so mr_bn_bwd(flags bit 3) skips its reduction pass over dx / x / y.  bn_x / bn_y: NHWC like dx (bn_y null = no fused
// ReLU); bn_mean / bn_rstd: the statistics mr_bn_fwd_train saved.  *produced = 1 when the sums were written; 0 when this
// geometry / dtype runs on a kernel without that epilogue (strided or > 32-tap dgrad, register-staged kernel): dx is complete
// either way and the caller then lets mr_bn_bwd reduce itself.
int mr_conv2d_dgrad_bnb(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, const void* bn_x,
                        const void* bn_y, const float* bn_mean, const float* bn_rstd, double* bn_sums, int* produced,
                        int Nimg, int H, int W, int Cin, int lddx, int Cout, int lddy, int R, int S, int sh, int sw, int ph,
                        int pw, int dh, int dw, int Ho, int Wo, hipStream_t stream) {
  MR_CHECK_ARG(bn_x && bn_mean && bn_rstd && bn_sums && produced, "mr_conv2d_dgrad_bnb: null BatchNorm operand");
  MR_CHECK_ARG(lddx == Cin, "mr_conv2d_dgrad_bnb: dx must be dense (lddx == Cin), like the BatchNorm's tensors");
  MR_CHECK_ARG(aligned16(bn_x) && (bn_y == nullptr || aligned16(bn_y)), "mr_conv2d_dgrad_bnb: bn_x / bn_y must be 16-byte aligned");
  g_epi_addend = addend;
  g_epi_stats = bn_sums;
  g_epi_stats_missed = false;
  g_epi_bnb = BnbRequest{bn_x, bn_y, bn_mean, bn_rstd};
  const int rc = mr_conv2d_dgrad(dtype, dy, w_crsk, dx, Nimg, H, W, Cin, lddx, Cout, lddy, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo,
                                 stream);
  *produced = g_epi_stats_missed ? 0 : 1;
  g_epi_addend = nullptr;
  g_epi_stats = nullptr;
  g_epi_stats_missed = false;
  g_epi_bnb = BnbRequest{nullptr, nullptr, nullptr, nullptr};
  return rc;
}

// dw[k,r,s,c] += sum_{n,ho,wo} dy[n,ho,wo,k] * x[n, ho*sh-ph+r*dh, wo*sw-pw+s*dw, c]   (f32, atomically accumulated)
int mr_conv2d_wgrad(int dtype, const void* dy, const void* x, float* dw_krsc, float* dbias, int Nimg, int H, int W,
                    int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw, int dh,
                    int dw, int Ho, int Wo, hipStream_t stream) {
  MR_CHECK_ARG(dtype == MR_F32 || dtype == MR_BF16, "mr_conv2d_wgrad: bad dtype %d", dtype);
  const int vec = dtype == MR_F32 ? 4 : 8;
  // Cout need not be a multiple of the vector width as long as the rows of dy are (lddy): the operand loads then read the
  // zero padding channels of the last vector, and only Cout rows of dw / entries of dbias are written
  MR_CHECK_ARG(Cin % vec == 0 && ldx % vec == 0 && lddy % vec == 0 && Cout > 0 && Cout <= lddy,
               "mr_conv2d_wgrad: Cin/ldx/lddy must be multiples of %d and Cout <= lddy", vec);
  MR_CHECK_ARG(aligned16(dy) && aligned16(x), "mr_conv2d_wgrad: dy and x must be 16-byte aligned");
  TnArgs a;
  a.A = dy; a.B = x; a.C = dw_krsc; a.P = Nimg * Ho * Wo; a.NA = Cout; a.NB = R * S * Cin; a.lda = lddy;
  a.ldb = 0; a.ldc = R * S * Cin; a.p_chunk = 0; a.row_perm_h = 0; a.colsum = dbias; a.rowtab = nullptr; a.grp = 1; a.ws = nullptr;
  ConvGeom g;
  fill_geom(g, 1, H, W, Cin, ldx, Ho, Wo, R, S, sh, sw, ph, pw, dh, dw);
  if (dtype == MR_F32) return launch_tn<float, 1>(a, g, stream);
  return launch_tn<bf16_t, 1>(a, g, stream);
}

// Same as mr_conv2d_wgrad with a caller-owned ROW TABLE of the gather (8 bytes per output pixel, see
// tn_rowtab_kernel): build != 0 fills it first (one small kernel), build == 0 trusts its contents -- the geometry of
// a layer never changes, so callers build it once and pass it to every later step.  bf16, R*S <= 32 only
// (otherwise, or with rowtab == null, this is mr_conv2d_wgrad).
static int conv2d_wgrad_tab_impl(int dtype, const void* dy, const void* x, float* dw_krsc, float* dbias, int Nimg, int H,
                                 int W, int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw,
                                 int dh, int dw, int Ho, int Wo, void* rowtab, int build, hipStream_t stream);
// build: bit 0 = fill the row table first; bit 1 = the launch may run CONCURRENTLY with other weight-gradient launches (a side
// stream): the shared split-reduction workspace is then left alone (f32 atomics only), see mr_set_tn_taps_workspace.
int mr_conv2d_wgrad_tab(int dtype, const void* dy, const void* x, float* dw_krsc, float* dbias, int Nimg, int H,
                        int W, int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw,
                        int dh, int dw, int Ho, int Wo, void* rowtab, int build, hipStream_t stream) {
  taps_set_concurrent((build & 2) != 0);
  const int rc = conv2d_wgrad_tab_impl(dtype, dy, x, dw_krsc, dbias, Nimg, H, W, Cin, ldx, Cout, lddy, R, S, sh, sw, ph, pw, dh,
                                       dw, Ho, Wo, rowtab, build & 1, stream);
  taps_set_concurrent(false);
  return rc;
}

static int conv2d_wgrad_tab_impl(int dtype, const void* dy, const void* x, float* dw_krsc, float* dbias, int Nimg, int H,
                                 int W, int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw,
                                 int dh, int dw, int Ho, int Wo, void* rowtab, int build, hipStream_t stream) {
  if (rowtab == nullptr || dtype != MR_BF16 || R * S > 32 || g_nt_variant != 2)
    return mr_conv2d_wgrad(dtype, dy, x, dw_krsc, dbias, Nimg, H, W, Cin, ldx, Cout, lddy, R, S, sh, sw, ph, pw, dh,
                           dw, Ho, Wo, stream);
  MR_CHECK_ARG(Cin % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && Cout > 0 && Cout <= lddy,
               "mr_conv2d_wgrad_tab: Cin/ldx/lddy must be multiples of 8 and Cout <= lddy");
  MR_CHECK_ARG(aligned16(dy) && aligned16(x) && ((uintptr_t)rowtab & 7) == 0, "mr_conv2d_wgrad_tab: alignment");
  if (g_tn_taps && taps_eligible(Nimg, H, W, Cin, ldx, Cout, lddy, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo,
                                 (long long)Nimg * Ho * Wo * 8)) {
    TapsProblem tp;
    tp.dy = dy; tp.x = x; tp.dw = dw_krsc; tp.dbias = dbias; tp.tab = (int*)rowtab; tp.build = build;
    tp.N = Nimg; tp.H = H; tp.W = W; tp.Cin = Cin; tp.ldx = ldx; tp.Cout = Cout; tp.lddy = lddy; tp.dil = dh;
    return launch_tn_taps(tp, g_tn_splits, stream);
  }
  TnArgs a;
  a.A = dy; a.B = x; a.C = dw_krsc; a.P = Nimg * Ho * Wo; a.NA = Cout; a.NB = R * S * Cin; a.lda = lddy;
  a.ldb = 0; a.ldc = R * S * Cin; a.p_chunk = 0; a.row_perm_h = 0; a.colsum = dbias; a.rowtab = (const int2*)rowtab; a.grp = 1; a.ws = nullptr;
  ConvGeom g;
  fill_geom(g, 1, H, W, Cin, ldx, Ho, Wo, R, S, sh, sw, ph, pw, dh, dw);
  if (build) {
    hipLaunchKernelGGL(tn_rowtab_kernel, dim3(cdiv(a.P, 256)), dim3(256), 0, stream, g, a.P, (int2*)rowtab);
    MR_CHECK_LAUNCH();
  }
  return launch_tn<bf16_t, 2>(a, g, stream);
}

}  // extern "C"
