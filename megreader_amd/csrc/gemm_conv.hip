// C-ABI entry points for the MFMA GEMM / implicit-GEMM convolution family.
// Replaces (on MI355X) the cuDNN / cuBLAS calls behind nn.Conv2d, nn.Linear and nn.LSTM's input
// projection on the reference's training path (reference: backbones/crnn.py:44-55,
// backbones/resnet.py:110-256, decoders/crnn.py:8-24).
#include "igemm_core.h"
#include "../../include/megreader_hip.h"

namespace mr {

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
  epi.vec_ok = ((ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0);
  const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
  hipLaunchKernelGGL((igemm_nt_kernel<T, BM, BN, AMODE, EpiStore<T>>), dim3(tiles), dim3(256), 0, stream, a, g,
                     epi);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// tile selection shared by every NT launch: 64-wide N tile for N <= 64, and 64-row tiles when
// 128-row tiles cannot fill the 256 CUs twice over
static void nt_tile(int M, int N, bool& m64, bool& n64) {
  n64 = N <= 64;
  const int blocks128 = cdiv(M, 128) * cdiv(N, n64 ? 64 : 128);
  m64 = blocks128 < 512;
}

template <typename T, int AMODE>
static int dispatch_nt_store(const NtArgs& a, const ConvGeom& g, void* C, long long ldc, const float* bias,
                             int relu, hipStream_t stream) {
  bool m64, n64;
  nt_tile(a.M, a.N, m64, n64);
  if (m64) {
    if (n64) return launch_nt_store<T, 64, 64, AMODE>(a, g, C, ldc, bias, relu, stream);
    return launch_nt_store<T, 64, 128, AMODE>(a, g, C, ldc, bias, relu, stream);
  }
  if (n64) return launch_nt_store<T, 128, 64, AMODE>(a, g, C, ldc, bias, relu, stream);
  return launch_nt_store<T, 128, 128, AMODE>(a, g, C, ldc, bias, relu, stream);
}

template <typename T, int BMODE>
static int launch_tn(TnArgs a, const ConvGeom& g, hipStream_t stream) {
  constexpr int BP = TnCfg<T>::BP;
  const int tiles = cdiv(a.NA, 128) * cdiv(a.NB, 128);
  int splits = 2048 / tiles;
  if (splits < 1) splits = 1;
  const int max_splits = cdiv(a.P, BP * 4);  // at least 4 p-steps per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.p_chunk = cdiv(cdiv(a.P, splits), BP) * BP;
  splits = cdiv(a.P, a.p_chunk);
  hipLaunchKernelGGL((igemm_tn_kernel<T, BMODE>), dim3(tiles, 1, splits), dim3(256), 0, stream, a, g);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace mr

using namespace mr;

extern "C" {

// Which NT tile (BM*1000 + BN) a problem of M rows x N columns is dispatched to (profiling / bench labels).
int mr_nt_tile_code(int M, int N) {
  bool m64, n64;
  nt_tile(M, N, m64, n64);
  return (m64 ? 64 : 128) * 1000 + (n64 ? 64 : 128);
}

int mr_gemm_nt(int dtype, const void* A, long long lda, const void* B, int ldb, void* C, long long ldc,
               const float* bias, int relu, int M, int N, int K, hipStream_t stream) {
  MR_CHECK_ARG(dtype == MR_F32 || dtype == MR_BF16, "mr_gemm_nt: bad dtype %d", dtype);
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(M > 0 && N > 0 && K >= 0, "mr_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  MR_CHECK_ARG(K % vec == 0 && lda % vec == 0 && ldb % vec == 0,
               "mr_gemm_nt: K/lda/ldb must be multiples of %d (K=%d lda=%lld ldb=%d)", vec, K, lda, ldb);
  MR_CHECK_ARG(aligned16(A) && aligned16(B), "mr_gemm_nt: A and B must be 16-byte aligned");
  NtArgs a;
  a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
  ConvGeom g = {};
  if (dtype == MR_F32) return dispatch_nt_store<float, 0>(a, g, C, ldc, bias, relu, stream);
  return dispatch_nt_store<bf16_t, 0>(a, g, C, ldc, bias, relu, stream);
}

int mr_gemm_tn(int dtype, const void* A, long long lda, const void* B, long long ldb, float* C, int ldc, int P,
               int NA, int NB, int row_perm_h, float* colsum, hipStream_t stream) {
  MR_CHECK_ARG(dtype == MR_F32 || dtype == MR_BF16, "mr_gemm_tn: bad dtype %d", dtype);
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(P > 0 && NA > 0 && NB > 0, "mr_gemm_tn: bad shape P=%d NA=%d NB=%d", P, NA, NB);
  MR_CHECK_ARG(NA % vec == 0 && NB % vec == 0 && lda % vec == 0 && ldb % vec == 0,
               "mr_gemm_tn: NA/NB/lda/ldb must be multiples of %d", vec);
  MR_CHECK_ARG(aligned16(A) && aligned16(B), "mr_gemm_tn: A and B must be 16-byte aligned");
  MR_CHECK_ARG(row_perm_h == 0 || NA % (4 * row_perm_h) == 0, "mr_gemm_tn: NA must be a multiple of 4*row_perm_h");
  TnArgs a;
  a.A = A; a.B = B; a.C = C; a.P = P; a.NA = NA; a.NB = NB; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.p_chunk = 0; a.row_perm_h = row_perm_h; a.colsum = colsum;
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
  a.A = x; a.B = w_krsc; a.M = Nimg * Ho * Wo; a.N = Cout; a.K = R * S * Cin; a.lda = 0; a.ldb = R * S * Cin;
  ConvGeom g;
  fill_geom(g, 1, H, W, Cin, ldx, Ho, Wo, R, S, sh, sw, ph, pw, dh, dw);
  if (R * S <= 32) {
    if (dtype == MR_F32) return dispatch_nt_store<float, 2>(a, g, y, ldy, bias, relu, stream);
    return dispatch_nt_store<bf16_t, 2>(a, g, y, ldy, bias, relu, stream);
  }
  if (dtype == MR_F32) return dispatch_nt_store<float, 1>(a, g, y, ldy, bias, relu, stream);
  return dispatch_nt_store<bf16_t, 1>(a, g, y, ldy, bias, relu, stream);
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
  a.A = dy; a.B = w_crsk; a.M = Nimg * H * W; a.N = Cin; a.K = R * S * Cout; a.lda = 0; a.ldb = R * S * Cout;
  ConvGeom g;
  fill_geom(g, 2, Ho, Wo, Cout, lddy, H, W, R, S, sh, sw, ph, pw, dh, dw);
  if (R * S <= 32 && sh == 1 && sw == 1) {
    if (dtype == MR_F32) return dispatch_nt_store<float, 2>(a, g, dx, lddx, nullptr, 0, stream);
    return dispatch_nt_store<bf16_t, 2>(a, g, dx, lddx, nullptr, 0, stream);
  }
  if (dtype == MR_F32) return dispatch_nt_store<float, 1>(a, g, dx, lddx, nullptr, 0, stream);
  return dispatch_nt_store<bf16_t, 1>(a, g, dx, lddx, nullptr, 0, stream);
}

// dw[k,r,s,c] += sum_{n,ho,wo} dy[n,ho,wo,k] * x[n, ho*sh-ph+r*dh, wo*sw-pw+s*dw, c]   (f32, atomically accumulated)
int mr_conv2d_wgrad(int dtype, const void* dy, const void* x, float* dw_krsc, float* dbias, int Nimg, int H, int W,
                    int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw, int dh,
                    int dw, int Ho, int Wo, hipStream_t stream) {
  MR_CHECK_ARG(dtype == MR_F32 || dtype == MR_BF16, "mr_conv2d_wgrad: bad dtype %d", dtype);
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(Cin % vec == 0 && ldx % vec == 0 && Cout % vec == 0 && lddy % vec == 0,
               "mr_conv2d_wgrad: Cin/ldx/Cout/lddy must be multiples of %d", vec);
  MR_CHECK_ARG(aligned16(dy) && aligned16(x), "mr_conv2d_wgrad: dy and x must be 16-byte aligned");
  TnArgs a;
  a.A = dy; a.B = x; a.C = dw_krsc; a.P = Nimg * Ho * Wo; a.NA = Cout; a.NB = R * S * Cin; a.lda = lddy;
  a.ldb = 0; a.ldc = R * S * Cin; a.p_chunk = 0; a.row_perm_h = 0; a.colsum = dbias;
  ConvGeom g;
  fill_geom(g, 1, H, W, Cin, ldx, Ho, Wo, R, S, sh, sw, ph, pw, dh, dw);
  if (dtype == MR_F32) return launch_tn<float, 1>(a, g, stream);
  return launch_tn<bf16_t, 1>(a, g, stream);
}

}  // extern "C"
