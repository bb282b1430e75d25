// All-taps weight-gradient kernel for 3x3 / stride 1 / "same" convolutions (bf16): interface between tn_taps.hip
// (kernel + launcher) and gemm_conv.hip (the mr_conv2d_wgrad_tab entry point that dispatches to it).
#pragma once
#include "common.h"

namespace mr {

// Geometry of one launch.  dw[k][r][s][c] += sum_{n,h,w} dy[n,h,w,k] * x[n, h+(r-1)*d, w+(s-1)*d, c]
struct TapsProblem {
  const void* dy;   // [N*H*W][lddy] bf16
  const void* x;    // [N*H*W][ldx] bf16
  float* dw;        // [Cout][3][3][Cin] f32, accumulated atomically
  float* dbias;     // optional [Cout]: += column sums of dy
  int* tab;         // caller-owned table, >= taps_table_ints(...) ints
  int build;        // != 0: fill the table first
  int N, H, W, Cin, ldx, Cout, lddy, dil;
};

// != 0 when the all-taps kernel can serve the geometry AND its stream table fits in `tab_bytes` bytes.
int taps_eligible(int N, int H, int W, int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph,
                  int pw, int dh, int dw, int Ho, int Wo, long long tab_bytes);
// Workspace of the in-launch group reduction (tickets + slabs; see TapArgs.grp): caller-owned device memory, zeroed
// once by the caller; launches that use it must be stream-ordered with respect to each other.
void taps_set_workspace(void* p, long long bytes);
void taps_get_workspace(void** p, long long* bytes);   // the current device's registered workspace (or null, 0)
void taps_set_concurrent(bool on);   // this host thread's next launches must not touch the shared workspace
#ifdef MR_ABLATION
int taps_set_abl(int mask);   // timing-only ablations (wrong results), see tn_taps.hip
#endif
int launch_tn_taps(const TapsProblem& p, int splits_override, hipStream_t stream);

}  // namespace mr
