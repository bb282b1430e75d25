// Bidirectional single-layer LSTM (PyTorch / cuDNN semantics, gate order i,f,g,o) as MFMA GEMMs with the
// gate non-linearities and cell update fused into the GEMM epilogue.  Replaces nn.LSTM(bidirectional=True)
// at reference decoders/crnn.py:13,21 (cuDNN RNN in the reference).
//
// Internal layout ("gate-interleaved"): every 4H-wide gate axis is permuted so that column 4*j+q holds
// gate q (i,f,g,o) of hidden unit j.  With the swapped-operand MFMA each lane then owns all four gates
// of one (batch row, hidden unit) pair, so the recurrence epilogue needs no cross-lane traffic.
//
// Buffers (T = compute dtype, row-major):
//   xproj  [Tn*N, 8H]   x_t W_ih^T + b_ih + b_hh for both directions (dir-major: [dir][4H]), gate-interleaved
//   out    [Tn, N, 2H]  h_t ([:, :, 0:H] forward, [:, :, H:2H] reverse)
//   cbuf   [Tn, N, 2H]  c_t (f32)
//   gates  [Tn, N, 8H]  post-activation gates (fwd) ; reused in place for pre-activation gradients (bwd)
//   whh    [2][4H, H]   recurrent weights, gate-interleaved rows     (step forward:  B operand)
//   whhT   [2][H, 4H]   their transposes                             (step backward: B operand)
#include "igemm_core.h"
#include "../../include/megreader_hip.h"

namespace mr {

template <typename T> struct EpiLstmFwd {
  const T* xproj;   // row t, dir offset applied: [N][8H] rows, + dir*4H
  const float* c_prev;  // [N][2H] + dir*H, or null for the first step
  float* c_out;     // [N][2H] + dir*H
  T* h_out;         // [N][2H] + dir*H
  T* gates_out;     // [N][8H] + dir*4H
  int Nb, H;
  // Operands of the gate math that do not depend on the recurrent GEMM: fetched BEFORE the k-loop so that their
  // memory round trip overlaps the GEMM instead of following it (the step kernels are pure latency chains).
  struct Pre { f32x4 x; float cp; };
  __device__ __forceinline__ Pre prefetch(int m, int n) const {
    Pre p;
    p.x = f32x4{0.f, 0.f, 0.f, 0.f};
    p.cp = 0.f;
    if (m < Nb && n < 4 * H) {
      p.x = load4(xproj + (long long)m * 8 * H + n);
      if (c_prev) p.cp = c_prev[(long long)m * 2 * H + (n >> 2)];
    }
    return p;
  }
  __device__ __forceinline__ void apply(int m, int n, f32x4 v, const Pre& p) const {
    if (m >= Nb || n >= 4 * H) return;
    const int j = n >> 2;
    const float ig = sigmoidf_(v[0] + p.x[0]);
    const float fg = sigmoidf_(v[1] + p.x[1]);
    const float gg = tanhf_(v[2] + p.x[2]);
    const float og = sigmoidf_(v[3] + p.x[3]);
    const float c = fg * p.cp + ig * gg;
    const float h = og * tanhf_(c);
    c_out[(long long)m * 2 * H + j] = c;
    h_out[(long long)m * 2 * H + j] = from_f32<T>(h);
    f32x4 g4 = {ig, fg, gg, og};
    store4(gates_out + (long long)m * 8 * H + n, g4);
  }
};

template <typename T> struct EpiLstmBwd {
  const T* dout;        // [N][2H] + dir*H   upstream gradient of h_t
  const T* gates;       // [N][8H] + dir*4H  post-activation gates of step t (read)
  T* dgates;            // [N][8H] + dir*4H  pre-activation gradients of step t (written; may alias gates)
  const float* c_t;     // [N][2H] + dir*H
  const float* c_prev;  // [N][2H] + dir*H or null (first forward step)
  float* dc;            // [N][2H] + dir*H   carried cell gradient (read unless first bwd step, then written)
  int first;            // first backward step: no carried dc
  int Nb, H;
  // everything the gate algebra reads besides the recurrent GEMM result (see EpiLstmFwd::Pre)
  struct Pre { f32x4 up, g[4], ct, dcv, cp; };
  __device__ __forceinline__ Pre prefetch(int m, int n) const {
    Pre p;
    p.up = p.ct = p.dcv = p.cp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) p.g[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m < Nb && n < H) {
      const long long r2 = (long long)m * 2 * H, r8 = (long long)m * 8 * H;
      p.up = load4(dout + r2 + n);
      p.ct = *(const f32x4*)(c_t + r2 + n);
      if (!first) p.dcv = *(const f32x4*)(dc + r2 + n);
      if (c_prev) p.cp = *(const f32x4*)(c_prev + r2 + n);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) p.g[jj] = load4(gates + r8 + 4 * (n + jj));
    }
    return p;
  }
  __device__ __forceinline__ void apply(int m, int n, f32x4 v, const Pre& p) const {
    if (m >= Nb || n >= H) return;
    const long long r2 = (long long)m * 2 * H, r8 = (long long)m * 8 * H;
    f32x4 dc_out;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = n + jj;
      const float ig = p.g[jj][0], fg = p.g[jj][1], gg = p.g[jj][2], og = p.g[jj][3];
      const float dh = v[jj] + p.up[jj];
      const float tc = tanhf_(p.ct[jj]);
      const float dcv = dh * og * (1.f - tc * tc) + p.dcv[jj];
      f32x4 d;
      d[0] = dcv * gg * ig * (1.f - ig);
      d[1] = dcv * p.cp[jj] * fg * (1.f - fg);
      d[2] = dcv * ig * (1.f - gg * gg);
      d[3] = dh * tc * og * (1.f - og);
      dc_out[jj] = dcv * fg;
      store4(dgates + r8 + 4 * j, d);
    }
    *(f32x4*)(dc + r2 + n) = dc_out;
  }
};

constexpr int LSTM_BM = 16;      // batch rows per workgroup: small tiles => many workgroups share the gate epilogue
constexpr int LSTM_BN_FWD = 64;  // gate columns per workgroup (forward: 4H columns, K = H)
constexpr int LSTM_BN_BWD = 32;  // hidden columns per workgroup (backward: H columns, K = 4H -- the deeper reduction
                                 // gets narrower tiles so that all CUs share the longer fragment fetch)

// BN > 0: direct-fragment body with BN-column tiles; BN == 0: LDS-staged split-K body (64-column tiles)
template <typename T, int BN, typename Epi>
__global__ __launch_bounds__(256) void lstm_step_kernel(NtArgs a0, NtArgs a1, Epi e0, Epi e1) {
  const NtArgs a = blockIdx.y == 0 ? a0 : a1;
  const Epi e = blockIdx.y == 0 ? e0 : e1;
  if constexpr (BN == 0)
    igemm_nt_ksplit_body<T, LSTM_BM, Epi>(a, e);
  else
    igemm_nt_kdirect_body<T, LSTM_BM, BN, Epi>(a, e);
}

// tuning knobs (mr_tuning.lstm_fwd_bn / lstm_bwd_bn): column-tile width of the step kernels, 0 = LDS-staged split-K body
#define g_lstm_fwd_bn MR_TUNE(lstm_fwd_bn)
#define g_lstm_bwd_bn MR_TUNE(lstm_bwd_bn)   // default LSTM_BN_BWD (tuning.hip)

template <typename T>
static int lstm_fwd_impl(const void* xproj_, const void* whh_, void* out_, float* cbuf, void* gates_, int Tn, int N,
                         int H, hipStream_t stream) {
  const T* xproj = (const T*)xproj_;
  const T* whh = (const T*)whh_;
  T* out = (T*)out_;
  T* gates = (T*)gates_;
  const long long s2 = (long long)N * 2 * H, s8 = (long long)N * 8 * H;
  const int bn = g_lstm_fwd_bn;
  const int tiles = cdiv(N, LSTM_BM) * cdiv(4 * H, bn ? bn : 64);
  for (int s = 0; s < Tn; ++s) {
    NtArgs a[2];
    EpiLstmFwd<T> e[2];
    for (int d = 0; d < 2; ++d) {
      const int t = d == 0 ? s : Tn - 1 - s;
      const int tp = d == 0 ? t - 1 : t + 1;
      a[d].A = s == 0 ? (const void*)out : (const void*)(out + tp * s2 + d * H);
      a[d].B = whh + (long long)d * 4 * H * H;
      a[d].M = N; a[d].N = 4 * H; a[d].K = s == 0 ? 0 : H; a[d].lda = 2 * H; a[d].ldb = H; a[d].zero = nullptr;
      e[d].xproj = xproj + t * s8 + d * 4 * H;
      e[d].c_prev = s == 0 ? nullptr : cbuf + tp * s2 + d * H;
      e[d].c_out = cbuf + t * s2 + d * H;
      e[d].h_out = out + t * s2 + d * H;
      e[d].gates_out = gates + t * s8 + d * 4 * H;
      e[d].Nb = N; e[d].H = H;
    }
    if (bn == 0)
      hipLaunchKernelGGL((lstm_step_kernel<T, 0, EpiLstmFwd<T>>), dim3(tiles, 2), dim3(256), 0, stream, a[0], a[1],
                         e[0], e[1]);
    else if (bn == 32)
      hipLaunchKernelGGL((lstm_step_kernel<T, 32, EpiLstmFwd<T>>), dim3(tiles, 2), dim3(256), 0, stream, a[0], a[1],
                         e[0], e[1]);
    else
      hipLaunchKernelGGL((lstm_step_kernel<T, 64, EpiLstmFwd<T>>), dim3(tiles, 2), dim3(256), 0, stream, a[0], a[1],
                         e[0], e[1]);
  }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

template <typename T>
static int lstm_bwd_impl(const void* dout_, const void* whhT_, const float* cbuf, void* gates_, float* dc, int Tn,
                         int N, int H, hipStream_t stream) {
  const T* dout = (const T*)dout_;
  const T* whhT = (const T*)whhT_;
  T* gates = (T*)gates_;
  const long long s2 = (long long)N * 2 * H, s8 = (long long)N * 8 * H;
  const int bn = g_lstm_bwd_bn;
  const int tiles = cdiv(N, LSTM_BM) * cdiv(H, bn ? bn : 64);
  for (int s = 0; s < Tn; ++s) {
    NtArgs a[2];
    EpiLstmBwd<T> e[2];
    for (int d = 0; d < 2; ++d) {
      // backward visits the steps in the reverse of the forward order of that direction
      const int t = d == 0 ? Tn - 1 - s : s;
      const int tn = d == 0 ? t + 1 : t - 1;  // step whose dgates feed the recurrent term
      const int tp = d == 0 ? t - 1 : t + 1;  // previous step in forward order (c_{prev})
      const bool has_prev = d == 0 ? t > 0 : t < Tn - 1;
      a[d].A = s == 0 ? (const void*)gates : (const void*)(gates + tn * s8 + d * 4 * H);
      a[d].B = whhT + (long long)d * 4 * H * H;
      a[d].M = N; a[d].N = H; a[d].K = s == 0 ? 0 : 4 * H; a[d].lda = 8 * H; a[d].ldb = 4 * H; a[d].zero = nullptr;
      e[d].dout = dout + t * s2 + d * H;
      e[d].gates = gates + t * s8 + d * 4 * H;
      e[d].dgates = gates + t * s8 + d * 4 * H;
      e[d].c_t = cbuf + t * s2 + d * H;
      e[d].c_prev = has_prev ? cbuf + tp * s2 + d * H : nullptr;
      e[d].dc = dc + d * H;
      e[d].first = s == 0;
      e[d].Nb = N; e[d].H = H;
    }
    if (bn == 0)
      hipLaunchKernelGGL((lstm_step_kernel<T, 0, EpiLstmBwd<T>>), dim3(tiles, 2), dim3(256), 0, stream, a[0], a[1],
                         e[0], e[1]);
    else if (bn == 16)
      hipLaunchKernelGGL((lstm_step_kernel<T, 16, EpiLstmBwd<T>>), dim3(tiles, 2), dim3(256), 0, stream, a[0], a[1],
                         e[0], e[1]);
    else if (bn == 32)
      hipLaunchKernelGGL((lstm_step_kernel<T, 32, EpiLstmBwd<T>>), dim3(tiles, 2), dim3(256), 0, stream, a[0], a[1],
                         e[0], e[1]);
    else
      hipLaunchKernelGGL((lstm_step_kernel<T, 64, EpiLstmBwd<T>>), dim3(tiles, 2), dim3(256), 0, stream, a[0], a[1],
                         e[0], e[1]);
  }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// lstm_persist.hip
bool lstm_persist_ok(int dtype, int T, int N, int H);
long long lstm_persist_ws(int dtype, int T, int N, int H);
int lstm_fwd_persist(const void* xproj, const void* whh, void* out, float* cbuf, void* gates, int T, int N, void* ws,
                     long long ws_bytes, hipStream_t stream);
int lstm_bwd_persist(const void* dout, const void* whhT, const float* cbuf, void* gates, int T, int N, void* ws,
                     long long ws_bytes, hipStream_t stream);
void lstm_set_bwd_debug(float* p);
#define g_lstm_persist (MR_TUNE(lstm_persist) != 0)   // 2 = persistent kernels without the XCD block map

}  // namespace mr

using namespace mr;

extern "C" {

// Tuning knob: column-tile width of the recurrence step kernels.  fwd_bn in {0, 32, 64}, bwd_bn in {0, 16, 32, 64};
// 0 selects the LDS-staged split-K body, other values the direct-fragment body.  Negative = leave unchanged.
// Debug hook (host only, not part of the product path): when non-null the persistent backward kernel also writes the
// recurrent term dh_rec [T, N, 2H] (f32) it reduced for every step.
int mr_lstm_debug_buffer(float* p) {
  lstm_set_bwd_debug(p);
  return MR_OK;
}

// Host-only switch: 1 (default) = persistent one-launch recurrence where applicable, 0 = per-step launches,
// 2 = persistent without the XCD-colocating block map (A/B: every hand-off then crosses XCDs through sc1 stores).
// Bytes of exchange workspace the persistent recurrence wants for this problem; 0 = it does not apply (f32 parity
// mode, H != 256, batch too large for co-residency) and mr_lstm_fwd/bwd run one launch per step.
long long mr_lstm_ws_bytes(int dtype, int T, int N, int H) {
  return g_lstm_persist ? lstm_persist_ws(dtype, T, N, H) : 0;
}

// Recurrent part of the forward pass (input projection done by mr_gemm_nt beforehand).
// ws / ws_bytes: exchange workspace of the persistent kernel (mr_lstm_ws_bytes; zeroed by the call itself) or
// null / 0 for the per-step launches.  ws_bytes < 0: the workspace is |ws_bytes| long and the caller hands it over ALREADY
// ZEROED (a slice of a buffer it zeroes once per training step): the per-call memset (one launch of ~5 us, four per CRNN
// step) is skipped.
int mr_lstm_fwd(int dtype, const void* xproj, const void* whh, void* out, float* cbuf, void* gates, int T, int N,
                int H, void* ws, long long ws_bytes, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(T > 0 && N > 0 && H > 0 && H % vec == 0, "mr_lstm_fwd: bad shape T=%d N=%d H=%d", T, N, H);
  if (ws && ws_bytes != 0 && g_lstm_persist && lstm_persist_ok(dtype, T, N, H))
    return lstm_fwd_persist(xproj, whh, out, cbuf, gates, T, N, ws, ws_bytes, stream);
  if (dtype == MR_F32) return lstm_fwd_impl<float>(xproj, whh, out, cbuf, gates, T, N, H, stream);
  if (dtype == MR_BF16) return lstm_fwd_impl<bf16_t>(xproj, whh, out, cbuf, gates, T, N, H, stream);
  mr::set_error("mr_lstm_fwd: bad dtype %d", dtype);
  return MR_ERR_DTYPE;
}

// Backward through time.  On return `gates` holds the pre-activation gradients (gate-interleaved) that feed
// the weight / input gradient GEMMs.  dc: scratch f32 [N, 2H].
int mr_lstm_bwd(int dtype, const void* dout, const void* whhT, const float* cbuf, void* gates, float* dc, int T,
                int N, int H, void* ws, long long ws_bytes, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(T > 0 && N > 0 && H > 0 && H % vec == 0 && H % 4 == 0, "mr_lstm_bwd: bad shape T=%d N=%d H=%d", T, N, H);
  if (ws && ws_bytes != 0 && g_lstm_persist && lstm_persist_ok(dtype, T, N, H))
    return lstm_bwd_persist(dout, whhT, cbuf, gates, T, N, ws, ws_bytes, stream);
  if (dtype == MR_F32) return lstm_bwd_impl<float>(dout, whhT, cbuf, gates, dc, T, N, H, stream);
  if (dtype == MR_BF16) return lstm_bwd_impl<bf16_t>(dout, whhT, cbuf, gates, dc, T, N, H, stream);
  mr::set_error("mr_lstm_bwd: bad dtype %d", dtype);
  return MR_ERR_DTYPE;
}

}  // extern "C"
