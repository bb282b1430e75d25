// Either side of the training step: the input pipeline and the evaluation decode, on the GPU.
// All HBM-bound byte / index work: one pass over the data, coalesced along the fastest output axis, wave-level
// ballots for the order-dependent parts.  Replaces (reference file:line):
//   mr_ctc_greedy_decode    structure/representers/ctc_representer.py:20-34      (python double loop per sample, step)
//   mr_ctc2d_greedy_decode  structure/representers/ctc_representer2d.py:27-51
//   mr_seq_measure          structure/measurers/sequence_recognition_measurer.py:66-72,101-112 (+ editdistance.eval)
//   mr_resize_normalize     data/processes/resize_image.py:29-38 (cv2.resize on float32, INTER_LINEAR) +
//                           data/processes/normalize_image.py:8-17 (-= RGB_MEAN, /= 255, HWC -> CHW)
//   mr_encode_labels        concern/charsets.py:37-58 (index / string_to_label) + make_recognition_label.py:11-24
#include "common.h"
#include "../../include/megreader_hip.h"

namespace mr {

enum { PD_F32 = 0, PD_BF16 = 1, PD_F64 = 2 };

template <typename T> __device__ __forceinline__ double ldval(const void* p, long long i);
template <> __device__ __forceinline__ double ldval<float>(const void* p, long long i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ double ldval<bf16_t>(const void* p, long long i) {
  return (float)((const bf16_t*)p)[i];
}
template <> __device__ __forceinline__ double ldval<double>(const void* p, long long i) { return ((const double*)p)[i]; }

// Collapse rule shared by the 1-D and 2-D decoders for a chunk of up to 64 consecutive steps held one per lane.
// c: this lane's arg-max class (lanes >= n_valid ignored).  carry: `previous` entering the chunk (wave-uniform).
// Returns the number of symbols emitted by the chunk and updates carry; writes out[base + rank].
__device__ __forceinline__ int collapse_chunk(int c, int n_valid, int blank, int unknown, int& carry, int* out, int base) {
  const int lane = threadIdx.x & 63;
  const bool in = lane < n_valid;
  // an `unknown` is skipped WITHOUT updating `previous`: drop them first, then the usual "differs from predecessor"
  const bool keep = in && c != unknown;
  const unsigned long long km = __ballot(keep);
  const unsigned long long below = km & ((1ull << lane) - 1ull);
  const int src = below ? 63 - __clzll((long long)below) : 0;
  const int pc = __shfl(c, src, 64);
  const int prev = below ? pc : carry;
  const bool emit = keep && c != prev && c != blank;
  const unsigned long long em = __ballot(emit);
  if (emit) out[base + __popcll(em & ((1ull << lane) - 1ull))] = c;
  if (km) {
    const int last = 63 - __clzll((long long)km);
    carry = __shfl(c, last, 64);
  }
  return __popcll(em);
}

// pred[n, c, t] at p + n*sn + c*sc + t*st (elements).  One wave per sample.
template <typename T>
__global__ __launch_bounds__(64) void ctc_greedy_decode_kernel(const void* pred, long long sn, long long sc, long long st,
                                                               int N, int C, int Tn, int blank, int unknown, int* out,
                                                               int* out_len) {
  const int n = blockIdx.x, lane = threadIdx.x;
  int* o = out + (long long)n * Tn;
  for (int t = lane; t < Tn; t += 64) o[t] = blank;
  int carry = blank, count = 0;
  for (int t0 = 0; t0 < Tn; t0 += 64) {
    const int t = t0 + lane;
    int best = 0;
    if (t < Tn) {
      double bv = ldval<T>(pred, n * sn + t * st);
      for (int c = 1; c < C; ++c) {
        const double v = ldval<T>(pred, n * sn + c * sc + t * st);
        if (v > bv) { bv = v; best = c; }   // strict: first index wins ties (torch.argmax)
      }
    }
    __builtin_amdgcn_wave_barrier();
    count += collapse_chunk(best, min(64, Tn - t0), blank, unknown, carry, o, count);
  }
  if (lane == 0 && out_len) out_len[n] = count;
}

// classify[n,c,h,w], mask[n,0,h,w] with explicit element strides.  One wave per (sample, 64 columns).
// The decode is sequential along w, so one wave walks all column chunks of its sample.
__global__ __launch_bounds__(64) void ctc2d_greedy_decode_kernel(const float* cl, long long cn, long long cc, long long ch,
                                                                 long long cw, const float* mk, long long mn,
                                                                 long long mh, long long mw, int N, int C, int H, int W,
                                                                 int blank, int unknown, int* out, int* out_len) {
  const int n = blockIdx.x, lane = threadIdx.x;
  int* o = out + (long long)n * W;
  for (int w = lane; w < W; w += 64) o[w] = blank;
  int carry = blank, count = 0;
  for (int w0 = 0; w0 < W; w0 += 64) {
    const int w = w0 + lane;
    int best_c = 0;
    if (w < W) {
      // row pick: argmax_h max_c (classify * mask), product in f32 as `heatmap = classify * mask` of the reference
      int best_h = 0;
      float best_row = 0.f;
      for (int h = 0; h < H; ++h) {
        const float m = mk[n * mn + h * mh + w * mw];
        float rmax = cl[n * cn + h * ch + w * cw] * m;
        for (int c = 1; c < C; ++c) rmax = fmaxf(rmax, cl[n * cn + c * cc + h * ch + w * cw] * m);
        if (h == 0 || rmax > best_row) { best_row = rmax; best_h = h; }
      }
      const float m = mk[n * mn + best_h * mh + w * mw];
      float bv = cl[n * cn + best_h * ch + w * cw] * m;
      for (int c = 1; c < C; ++c) {
        const float v = cl[n * cn + c * cc + best_h * ch + w * cw] * m;
        if (v > bv) { bv = v; best_c = c; }
      }
    }
    __builtin_amdgcn_wave_barrier();
    count += collapse_chunk(best_c, min(64, W - w0), blank, unknown, carry, o, count);
  }
  if (lane == 0 && out_len) out_len[n] = count;
}

// Accuracy and normalised edit distance of id sequences.  One wave per sample; sequences (after dropping blank /
// unknown ids) must be <= 63 symbols.  fold: optional id -> canonical id table (case folding: `.upper()`).
__global__ __launch_bounds__(64) void seq_measure_kernel(const int* labels, int S, const int* preds, int S2, int N,
                                                         int blank, int unknown, const int* fold, int* acc, int* ed,
                                                         int* lab_len, double* score) {
  const int n = blockIdx.x, lane = threadIdx.x;
  __shared__ int a_s[64], b_s[64];
  // compact both sequences (concern/charsets.py:60-62 label_to_string drops blank and unknown)
  int la = 0, lb = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const int* src = pass == 0 ? labels + (long long)n * S : preds + (long long)n * S2;
    const int len = pass == 0 ? S : S2;
    int* dst = pass == 0 ? a_s : b_s;
    int cnt = 0;
    for (int p0 = 0; p0 < len; p0 += 64) {
      const int p = p0 + lane;
      int v = p < len ? src[p] : blank;
      const bool keep = p < len && v != blank && v != unknown;
      if (keep && fold) v = fold[v];
      const unsigned long long km = __ballot(keep);
      const int pos = cnt + __popcll(km & ((1ull << lane) - 1ull));
      if (keep && pos < 64) dst[pos] = v;
      cnt += __popcll(km);
    }
    if (pass == 0) la = cnt; else lb = cnt;
  }
  __syncthreads();
  const bool too_long = la > 63 || lb > 63;
  const int La = min(la, 63), Lb = min(lb, 63);
  bool eq = la == lb;
  if (lane < La && lane < Lb) eq = eq && a_s[lane] == b_s[lane];
  const bool same = __all(eq);
  // anti-diagonal Levenshtein: lane j walks column j; cell (i, j) is computed at step d = i + j
  const int j = lane;
  const int bj = (j >= 1 && j <= Lb) ? b_s[j - 1] : -1;
  int v1 = 0, v2 = 0;   // this lane's cell values at steps d-1 and d-2
  int result = 0;
  for (int d = 0; d <= La + Lb; ++d) {
    const int left = __shfl_up(v1, 1, 64);    // D[i][j-1]   (lane j-1, step d-1)
    const int diag = __shfl_up(v2, 1, 64);    // D[i-1][j-1] (lane j-1, step d-2)
    const int i = d - j;
    int val = v1;
    const bool active = j <= Lb && i >= 0 && i <= La;
    if (active) {
      if (i == 0) val = j;
      else if (j == 0) val = i;
      else {
        const int cost = a_s[i - 1] != bj;
        val = min(min(v1 + 1, left + 1), diag + cost);   // v1 = D[i-1][j] (this lane, step d-1)
      }
    }
    v2 = v1;
    v1 = active ? val : v1;
    if (active && i == La && j == Lb) result = val;
  }
  result = __shfl(result, Lb, 64);
  if (lane == 0) {
    acc[n] = same ? 1 : 0;
    ed[n] = too_long ? -1 : result;
    lab_len[n] = la;
    // sequence_recognition_measurer.py:106-111 in the same IEEE double operations as the python expression
    score[n] = la == 0 ? 0.0 : (double)(1 - (double)min(la, result) * 1.0 / (double)la);
  }
}

struct ImgDesc {        // one source image (uint8, HWC, 3 channels)
  long long offset;     // byte offset into the packed source buffer
  int h, w, pitch;      // rows, columns, bytes per row
  int dst_w;            // columns of the resized image inside the canvas (== canvas width for mode "resize")
  double scale_x, scale_y;  // cv2: scale = 1. / ((double)dsize / ssize), evaluated by the host (one IEEE division
                            // sequence for the kernel and its oracle; not ssize / dsize, which rounds differently)
};

// dst[n, c, y, x] = ((double)resize(src_n)[y, x, c] - mean[c]) -> f32, / 255.f ; canvas columns >= dst_w hold the
// normalised value of a zero pixel (mode "pad": resize_image.py:48-53 pastes into a zero canvas BEFORE normalising).
// cv2.resize float32 INTER_LINEAR: fx = (float)((x + 0.5) * scale - 0.5) with scale in double, taps clamped with the
// weight of the out-of-range tap forced to 0, horizontal pass then vertical pass, plain float mul/add (no fma).
__global__ __launch_bounds__(256) void resize_normalize_kernel(const unsigned char* src, const ImgDesc* desc, int N,
                                                               int Hd, int Wd, double m0, double m1, double m2,
                                                               float* dst) {
  // bit-exactness against the host arithmetic: every product and sum below is rounded separately (hipcc contracts
  // a*b + c*d into an fma by default, and HIP's __fmul_rn / __fadd_rn are plain operators that get contracted too)
#pragma clang fp contract(off)
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)Hd * Wd;
  if (gid >= per * N) return;
  const int n = (int)(gid / per);
  const int y = (int)((gid % per) / Wd), x = (int)(gid % Wd);
  const ImgDesc d = desc[n];
  float v[3] = {0.f, 0.f, 0.f};
  if (x < d.dst_w) {
    const unsigned char* s = src + d.offset;
    if (d.h == Hd && d.w == d.dst_w) {   // cv2.resize returns a copy when the size already matches
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (float)s[(long long)y * d.pitch + 3 * x + c];
    } else {
      const double sx_scale = d.scale_x, sy_scale = d.scale_y;
      float fx = (float)((x + 0.5) * sx_scale - 0.5);   // rounded product, then rounded sum (contraction is off)
      int sx = (int)floorf(fx);
      fx -= sx;
      if (sx < 0) { sx = 0; fx = 0.f; }
      if (sx >= d.w - 1) { sx = d.w - 1; fx = 0.f; }
      float fy = (float)((y + 0.5) * sy_scale - 0.5);
      int sy = (int)floorf(fy);
      fy -= sy;
      if (sy < 0) { sy = 0; fy = 0.f; }
      if (sy >= d.h - 1) { sy = d.h - 1; fy = 0.f; }
      const int sx1 = min(sx + 1, d.w - 1), sy1 = min(sy + 1, d.h - 1);
      const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
      const unsigned char* r0 = s + (long long)sy * d.pitch;
      const unsigned char* r1 = s + (long long)sy1 * d.pitch;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float top = (float)r0[3 * sx + c] * a0 + (float)r0[3 * sx1 + c] * a1;
        const float bot = (float)r1[3 * sx + c] * a0 + (float)r1[3 * sx1 + c] * a1;
        v[c] = top * b0 + bot * b1;
      }
    }
  }
  const double mean[3] = {m0, m1, m2};
  float* o = dst + (long long)n * 3 * per + (long long)y * Wd + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c * per] = (float)((double)v[c] - mean[c]) / 255.f;
}

// label[n, p] = id of codepoint cp[off[n] + p] (binary search in the sorted table; missing -> unknown), 0 beyond the
// string; length[n] = min(len, max_size).  charsets.py:52-58 + make_recognition_label.py:21-24.
__global__ __launch_bounds__(256) void encode_labels_kernel(const int* cps, const long long* offs, int N, int max_size,
                                                            const int* tab_cp, const int* tab_id, int ntab,
                                                            int unknown, int* label, int* length) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= N * max_size) return;
  const int n = gid / max_size, p = gid % max_size;
  const long long lo = offs[n], hi = offs[n + 1];
  int id = 0;
  if (lo + p < hi) {
    const int cp = cps[lo + p];
    int a = 0, b = ntab - 1;
    id = unknown;
    while (a <= b) {
      const int mid = (a + b) >> 1;
      const int t = tab_cp[mid];
      if (t == cp) { id = tab_id[mid]; break; }
      if (t < cp) a = mid + 1; else b = mid - 1;
    }
  }
  label[gid] = id;
  if (p == 0) length[n] = (int)min((long long)max_size, hi - lo);
}

}  // namespace mr

using namespace mr;

extern "C" {

int mr_sizeof_img_desc(void) { return (int)sizeof(ImgDesc); }

int mr_ctc_greedy_decode(int dtype, const void* pred, long long sn, long long sc, long long st, int N, int C, int T,
                         int blank, int unknown, int* out, int* out_len, hipStream_t stream) {
  MR_CHECK_ARG(N >= 0 && C > 0 && T > 0, "mr_ctc_greedy_decode: bad shape N=%d C=%d T=%d", N, C, T);
  if (N == 0) return MR_OK;
  if (dtype == PD_F32)
    hipLaunchKernelGGL(ctc_greedy_decode_kernel<float>, dim3(N), dim3(64), 0, stream, pred, sn, sc, st, N, C, T, blank,
                       unknown, out, out_len);
  else if (dtype == PD_BF16)
    hipLaunchKernelGGL(ctc_greedy_decode_kernel<bf16_t>, dim3(N), dim3(64), 0, stream, pred, sn, sc, st, N, C, T,
                       blank, unknown, out, out_len);
  else if (dtype == PD_F64)
    hipLaunchKernelGGL(ctc_greedy_decode_kernel<double>, dim3(N), dim3(64), 0, stream, pred, sn, sc, st, N, C, T,
                       blank, unknown, out, out_len);
  else {
    mr::set_error("mr_ctc_greedy_decode: bad dtype %d", dtype);
    return MR_ERR_DTYPE;
  }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_ctc2d_greedy_decode(const float* classify, long long cn, long long cc, long long ch, long long cw,
                           const float* mask, long long mn, long long mh, long long mw, int N, int C, int H, int W,
                           int blank, int unknown, int* out, int* out_len, hipStream_t stream) {
  MR_CHECK_ARG(N >= 0 && C > 0 && H > 0 && W > 0, "mr_ctc2d_greedy_decode: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
  if (N == 0) return MR_OK;
  hipLaunchKernelGGL(ctc2d_greedy_decode_kernel, dim3(N), dim3(64), 0, stream, classify, cn, cc, ch, cw, mask, mn, mh,
                     mw, N, C, H, W, blank, unknown, out, out_len);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_seq_measure(const int* labels, int S, const int* preds, int S2, int N, int blank, int unknown, const int* fold,
                   int* acc, int* ed, int* label_len, double* score, hipStream_t stream) {
  MR_CHECK_ARG(N >= 0 && S > 0 && S2 > 0, "mr_seq_measure: bad shape N=%d S=%d S2=%d", N, S, S2);
  if (N == 0) return MR_OK;
  hipLaunchKernelGGL(seq_measure_kernel, dim3(N), dim3(64), 0, stream, labels, S, preds, S2, N, blank, unknown, fold,
                     acc, ed, label_len, score);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_resize_normalize(const unsigned char* src, const void* desc, int N, int H, int W, double mean0, double mean1,
                        double mean2, float* dst, hipStream_t stream) {
  MR_CHECK_ARG(N >= 0 && H > 0 && W > 0, "mr_resize_normalize: bad shape N=%d H=%d W=%d", N, H, W);
  if (N == 0) return MR_OK;
  const long long total = (long long)N * H * W;
  hipLaunchKernelGGL(resize_normalize_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, stream, src,
                     (const ImgDesc*)desc, N, H, W, mean0, mean1, mean2, dst);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_encode_labels(const int* codepoints, const long long* offsets, int N, int max_size, const int* table_cp,
                     const int* table_id, int ntab, int unknown, int* label, int* length, hipStream_t stream) {
  MR_CHECK_ARG(N >= 0 && max_size > 0 && ntab >= 0, "mr_encode_labels: bad shape N=%d max_size=%d", N, max_size);
  if (N == 0) return MR_OK;
  hipLaunchKernelGGL(encode_labels_kernel, dim3(cdiv(N * max_size, 256)), dim3(256), 0, stream, codepoints, offsets, N,
                     max_size, table_cp, table_id, ntab, unknown, label, length);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
