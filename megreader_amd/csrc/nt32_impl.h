// Per-shape launcher of the ping-pong 32x32x16 NT kernel: included by nt32_s<N>.hip with MR_NT32_SHAPE defined (one translation
// unit per tile shape: the instantiations are the slowest compiles of the library).
#include "nt32.h"
#include "igemm_nt32.h"

#if MR_NT32_SHAPE == 1      // 256x256: 2x4 waves of 128x64
#define S_WM 2
#define S_WN 4
#define S_TM 4
#define S_TN 2
#define S_PH 2
#elif MR_NT32_SHAPE == 2    // 288x256: 1x8 waves of 288x32
#define S_WM 1
#define S_WN 8
#define S_TM 9
#define S_TN 1
#define S_PH 4
#elif MR_NT32_SHAPE == 3    // 256x128: 2x4 waves of 128x32
#define S_WM 2
#define S_WN 4
#define S_TM 4
#define S_TN 1
#define S_PH 1
#elif MR_NT32_SHAPE == 4    // 128x256: 2x4 waves of 64x64
#define S_WM 2
#define S_WN 4
#define S_TM 2
#define S_TN 2
#define S_PH 1
#endif

namespace mr {

template <int AMODE, int PH, int OPT, int ABL = 0>
static int launch_one(const NtArgs& a, const ConvGeom& g, const EpiStore<bf16_t>& epi, hipStream_t stream) {
  constexpr int BM = S_WM * S_TM * 32, BN = S_WN * S_TN * 32;
  constexpr size_t lds_stage = 2 * (size_t)(BM + BN) * 128, lds_out = (size_t)BM * BN * 2;   // stage buffers; output tile (epilogue)
  constexpr size_t lds = lds_stage > lds_out ? lds_stage : lds_out;
  auto kern = igemm_nt32_kernel<S_WM, S_WN, S_TM, S_TN, AMODE, EpiStore<bf16_t>, PH, OPT, ABL>;
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
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a, g, epi);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

template <int AMODE>
static int launch_variant(int variant, const NtArgs& a, const ConvGeom& g, const EpiStore<bf16_t>& epi, hipStream_t stream) {
  switch (variant) {
#ifdef MR_NT32_SWEEP   // tools build: schedule variants 10 * PH + OPT
    case 20: return launch_one<AMODE, 2, 0>(a, g, epi, stream);
    case 21: return launch_one<AMODE, 2, 1>(a, g, epi, stream);
    case 28: return launch_one<AMODE, 2, 8>(a, g, epi, stream);
    case 26: return launch_one<AMODE, 2, 16>(a, g, epi, stream);
    case 48: return launch_one<AMODE, 4, 8>(a, g, epi, stream);
    case 40: return launch_one<AMODE, 4, 0>(a, g, epi, stream);
#endif
#ifdef MR_ABLATION     // timing-only ablations (wrong results; tools build): no LDS-DMA in the loop / no fragment reads / neither
    case 91: return launch_one<AMODE, S_PH, 0, 1>(a, g, epi, stream);
    case 92: return launch_one<AMODE, S_PH, 0, 2>(a, g, epi, stream);
    case 93: return launch_one<AMODE, S_PH, 0, 3>(a, g, epi, stream);
    case 94: return launch_one<AMODE, S_PH, 0, 4>(a, g, epi, stream);
    case 95: return launch_one<AMODE, S_PH, 0, 7>(a, g, epi, stream);
#endif
    default: return launch_one<AMODE, S_PH, 0>(a, g, epi, stream);
  }
}

#define MR_CAT2(a, b) a##b
#define MR_CAT(a, b) MR_CAT2(a, b)
int MR_CAT(launch_nt32_s, MR_NT32_SHAPE)(int variant, int amode, const NtArgs& a, const ConvGeom& g, const EpiStore<bf16_t>& epi,
                                         hipStream_t stream) {
  if (amode == 0) return launch_variant<0>(variant, a, g, epi, stream);
  if (amode == 2) return launch_variant<2>(variant, a, g, epi, stream);
  set_error("launch_nt32: amode %d", amode);
  return MR_ERR_ARG;
}

}  // namespace mr
