// Interface between nt32.hip (ping-pong 32x32x16 NT kernel, igemm_nt32.h) and gemm_conv.hip (dispatch_nt_store).
#pragma once
#include "igemm_core.h"

namespace mr {

// shape: 1 = 256x256 (2x4 waves of 128x64), 2 = 288x256 (1x8 waves of 288x32), 3 = 256x128 (2x4 waves of 128x32),
//        4 = 128x256 (2x4 waves of 64x64).  variant = mr_tuning.nt_m32_opt (10 * PH + OPT; 0 = the default of the shape).
// amode: 0 dense A, 2 fast conv gather.  a.zero is not used by this kernel.
int nt32_tile(int shape, int* bm, int* bn);
int launch_nt32(int shape, int variant, int amode, const NtArgs& a, const ConvGeom& g, const EpiStore<bf16_t>& epi,
                hipStream_t stream);

}  // namespace mr
