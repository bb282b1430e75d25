// HBM-bound helper kernels: layout conversion, ReLU backward, column sums, weight preparation
// (fp32 master -> compute-dtype operand images), fused Adam.  All 16-byte vectorised where the
// tensors are large (activations); weight-sized tensors use simple grid-stride loops.
#include <mutex>
#include <vector>
#include "common.h"
#include "../../include/megreader_hip.h"
#include <stdarg.h>

namespace mr {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline int grid_for(long long n, int block, int max_blocks = 8192) {
  long long b = (n + block - 1) / block;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------- NCHW f32 -> NHWC(T), channel padded
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int H, int W,
                                    int Cpad) {
  const long long total = (long long)N * H * W * Cpad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const long long p = i / Cpad;
    const int w = (int)(p % W);
    const long long q = p / W;
    const int h = (int)(q % H);
    const int n = (int)(q / H);
    float v = 0.f;
    if (c < C) v = src[(((long long)n * C + c) * H + h) * W + w];
    dst[i] = from_f32<T>(v);
  }
}

// NHWC(T) [N,H,W,ld] (first C channels) -> NCHW f32
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                    int ld) {
  const long long total = (long long)N * C * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long long q = i / W;
    const int h = (int)(q % H);
    const long long q2 = q / H;
    const int c = (int)(q2 % C);
    const int n = (int)(q2 / C);
    dst[i] = to_f32(src[(((long long)n * H + h) * W + w) * ld + c]);
  }
}

// ---------------------------------------------------------------- cast
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = from_f32<D>(to_f32(src[i]));
}

// ---------------------------------------------------------------- relu backward: dx = y > 0 ? dy : 0
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long long n) {
  constexpr int VEC = VecOf<T>::N;
  const long long nv = n / VEC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nv;
       i += (long long)gridDim.x * blockDim.x) {
    uint4 a = ((const uint4*)dy)[i];
    uint4 b = ((const uint4*)y)[i];
    T* pa = (T*)&a;
    const T* pb = (const T*)&b;
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      if (!(to_f32(pb[j]) > 0.f)) pa[j] = from_f32<T>(0.f);
    ((uint4*)dx)[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = nv * VEC; i < n; ++i) dx[i] = to_f32(y[i]) > 0.f ? dy[i] : from_f32<T>(0.f);
}

// ---------------------------------------------------------------- out = a + b (optionally relu)
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long n,
                           int relu) {
  constexpr int VEC = VecOf<T>::N;
  const long long nv = n / VEC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nv;
       i += (long long)gridDim.x * blockDim.x) {
    uint4 x = ((const uint4*)a)[i];
    uint4 y = ((const uint4*)b)[i];
    T* px = (T*)&x;
    const T* py = (const T*)&y;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = to_f32(px[j]) + to_f32(py[j]);
      if (relu) v = fmaxf(v, 0.f);
      px[j] = from_f32<T>(v);
    }
    ((uint4*)o)[i] = x;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = nv * VEC; i < n; ++i) {
      float v = to_f32(a[i]) + to_f32(b[i]);
      if (relu) v = fmaxf(v, 0.f);
      o[i] = from_f32<T>(v);
    }
}

// ---------------------------------------------------------------- column sums: out[perm(c)] += sum_p x[p, c]
// grid = (ceil(C/64), row_splits); block 256 = 4 waves striding the rows; lane = column.
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int P, int C, long long ld,
                              int rows_per_block, int perm_h) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int p0 = blockIdx.y * rows_per_block;
  const int p1 = min(P, p0 + rows_per_block);
  float s = 0.f;
  if (c < C)
    for (int p = p0 + wave; p < p1; p += 4) s += to_f32(x[(long long)p * ld + c]);
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < C) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    int oc = c;
    if (perm_h > 0) {
      const int h4 = 4 * perm_h;
      const int blk = c / h4, rin = c - blk * h4;
      oc = blk * h4 + (rin & 3) * perm_h + (rin >> 2);
    }
    atomicAdd(out + oc, s);
  }
}

// ---------------------------------------------------------------- [A,B,C] -> [B,A,C]
template <typename T>
__global__ void permute_021_kernel(const T* __restrict__ src, T* __restrict__ dst, int A, int B, int C) {
  constexpr int VEC = VecOf<T>::N;
  const int cv = C / VEC;
  const long long total = (long long)A * B * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    const long long q = i / cv;
    const int b = (int)(q % B);
    const int a = (int)(q / B);
    ((uint4*)dst)[((long long)b * A + a) * cv + c] = ((const uint4*)src)[i];
  }
}

// ---------------------------------------------------------------- conv weight prep
// src: f32, logical [K][C][R][S] with arbitrary element strides (sk, sc, sr, ss)
// dst_krsc: T [K][R][S][Cpad] (zero padded channels), dst_crsk: T [Cpad'..] = [C][R][S][K]  (either may be null)
template <typename T>
__global__ void prep_conv_weight_kernel(const float* __restrict__ src, long long sk, long long sc, long long sr,
                                        long long ss, T* __restrict__ dst_krsc, T* __restrict__ dst_crsk, int K,
                                        int C, int R, int S, int Cpad, int ldk) {
  const long long total = (long long)K * R * S * Cpad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    long long q = i / Cpad;
    const int s = (int)(q % S);
    q /= S;
    const int r = (int)(q % R);
    const int k = (int)(q / R);
    float v = 0.f;
    if (c < C) v = src[k * sk + c * sc + r * sr + s * ss];
    if (dst_krsc) dst_krsc[i] = from_f32<T>(v);
    if (dst_crsk && c < C) dst_crsk[(((long long)c * R + r) * S + s) * ldk + k] = from_f32<T>(v);
  }
}

// ---------------------------------------------------------------- matrix prep (linear / LSTM weights)
// src f32 [R][C] row-major.  dst (T) row r' = perm(r) where, if perm_h > 0, r = q*perm_h + j -> r' = 4*j + q
// (PyTorch gate-major i,f,g,o rows -> gate-interleaved rows).  dst_n: [R][ldn] normal, dst_t: [C][ldt] transposed.
template <typename T>
__global__ void prep_matrix_kernel(const float* __restrict__ src, int lds, T* __restrict__ dst_n, int ldn,
                                   T* __restrict__ dst_t, int ldt, int R, int C, int perm_h) {
  const long long total = (long long)R * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int r = (int)(i / C);
    int rp = r;
    if (perm_h > 0) {
      const int h4 = 4 * perm_h;
      const int blk = r / h4, rin = r - blk * h4;
      rp = blk * h4 + 4 * (rin % perm_h) + rin / perm_h;
    }
    const T v = from_f32<T>(src[(long long)r * lds + c]);
    if (dst_n) dst_n[(long long)rp * ldn + c] = v;
    if (dst_t) dst_t[(long long)c * ldt + rp] = v;
  }
}

// dst[perm(r)] = a[r] + b[r]   (LSTM bias sum, gate-interleaved)
__global__ void prep_bias_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst,
                                 int R, int perm_h) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  int rp = r;
  if (perm_h > 0) {
    const int h4 = 4 * perm_h;
    const int blk = r / h4, rin = r - blk * h4;
    rp = blk * h4 + 4 * (rin % perm_h) + rin / perm_h;
  }
  dst[rp] = a[r] + (b ? b[r] : 0.f);
}

// ---------------------------------------------------------------- batched weight prep / multi-segment accumulate
// One launch for every prepared operand image of a model (see mr_prep_batch in the header).  Work unit = one 64x64
// tile of a job's logical matrix (conv: rows k, columns (r,s,c) with c padded to Cpad; matrix: rows in DESTINATION
// (gate-interleaved) order, columns c): the tile is loaded once (all 16 loads of a thread in flight together), cast,
// written row-wise to the normal image and -- through LDS -- column-wise to the transposed image, so that both
// images are written in full 128-byte runs (the transposed one used to be 2-byte scatter writes).
template <typename T>
__global__ __launch_bounds__(256) void prep_batch_kernel(const mr_prep_job* __restrict__ jobs, int njobs, float* tick) {
  __shared__ int starts[1024];
  if (tick && blockIdx.x == 0 && threadIdx.x == 0) tick[5] += 1.f;   // the optimizer's step counter (mr_adam_step)
  __shared__ T tile[64][66];
  for (int t = threadIdx.x; t < njobs; t += blockDim.x) starts[t] = jobs[t].block_start;
  __syncthreads();
  int lo = 0, hi = njobs - 1;  // last job whose block_start <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const mr_prep_job j = jobs[lo];
  const int lb = (int)blockIdx.x - j.block_start;
  const int tid = threadIdx.x;
  if (j.kind == MR_PREP_BIAS) {  // f32 dst[perm(r)] = a[r] + b[r]
    const int R = j.d0, perm_h = j.perm_h;
    float* dst = (float*)j.dst_a;
    for (int r = lb * 4096 + tid; r < R && r < (lb + 1) * 4096; r += 256) {
      int rp = r;
      if (perm_h > 0) {
        const int h4 = 4 * perm_h;
        const int blk = r / h4, rin = r - blk * h4;
        rp = blk * h4 + 4 * (rin % perm_h) + rin / perm_h;
      }
      dst[rp] = j.src[r] + (j.src2 ? j.src2[r] : 0.f);
    }
    return;
  }
  if (j.kind == MR_PREP_STEM) {  // bf16-style packed stem filter bank [64][32] (see csrc/stem.hip)
    const int CIN = j.d1, KT = 9 * CIN;
    T* dst = (T*)j.dst_a;
    for (int i = tid; i < 64 * 32; i += 256) {
      const int ch = i >> 5, k = i & 31;
      const int dr = k / (3 * CIN), rem = k - dr * 3 * CIN, ds = rem / CIN, c = rem - ds * CIN;
      dst[i] = from_f32<T>(k < KT ? j.src[ch * j.s0 + c * j.s1 + dr * j.s2 + ds * j.s3] : 0.f);
    }
    return;
  }
  const bool conv = j.kind == MR_PREP_CONV;
  const int rows = j.d0;
  const int cols = conv ? j.d2 * j.d3 * j.pad : j.d1;
  const int tiles_c = (cols + 63) / 64;
  const int r0 = (lb / tiles_c) * 64, c0 = (lb % tiles_c) * 64;

  // ---- phase 1: load + cast; row-wise image
  const int col = tid & 63, rq = tid >> 6;
  const int colg = c0 + col;
  long long src_col = 0;   // source offset contributed by the column
  bool col_ok = colg < cols;
  if (conv) {
    const int c = colg % j.pad, rs = colg / j.pad;
    const int r = rs / j.d3, s = rs - r * j.d3;
    src_col = c * j.s1 + r * j.s2 + s * j.s3;
    col_ok = col_ok && c < j.d1;  // padded channels are zeros
  } else {
    src_col = colg;
  }
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rg = r0 + rq + 4 * i;  // destination row
    long long src_row;
    if (conv) {
      src_row = (long long)rg * j.s0;
    } else {
      int r = rg;
      if (j.perm_h > 0) {  // inverse of r = q*H + jj -> rp = 4*jj + q
        const int h4 = 4 * j.perm_h;
        const int blk = rg / h4, rin = rg - blk * h4;
        r = blk * h4 + (rin & 3) * j.perm_h + (rin >> 2);
      }
      src_row = (long long)r * j.s0;
    }
    v[i] = (col_ok && rg < rows) ? j.src[src_row + src_col] : 0.f;
  }
  T* dst_a = (T*)j.dst_a;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rl = rq + 4 * i, rg = r0 + rl;
    const T t = from_f32<T>(v[i]);
    tile[rl][col] = t;
    if (dst_a && rg < rows && colg < cols) dst_a[(long long)rg * (conv ? cols : j.pad) + colg] = t;
  }
  T* dst_b = (T*)j.dst_b;
  if (!dst_b) return;
  __syncthreads();
  // ---- phase 2: column-wise (transposed) image: thread = (tile column, 16-row segment)
  const int tc = tid >> 2, seg = tid & 3;
  const int cg = c0 + tc;
  if (cg >= cols) return;
  long long trow;
  if (conv) {
    const int c = cg % j.pad, rs = cg / j.pad;
    if (c >= j.d1) return;
    trow = (long long)c * j.d2 * j.d3 + rs;  // (c*R + r)*S + s
  } else {
    trow = cg;
  }
  T* out = dst_b + trow * j.ld_b + r0 + seg * 16;
  if (r0 + 64 <= rows && (j.ld_b % (16 / (int)sizeof(T))) == 0) {
    T buf[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) buf[e] = tile[seg * 16 + e][tc];
    uint4* o4 = (uint4*)out;
    const uint4* b4 = (const uint4*)buf;
#pragma unroll
    for (int q = 0; q < (int)(16 * sizeof(T) / 16); ++q) o4[q] = b4[q];
  } else {
    for (int e = 0; e < 16; ++e)
      if (r0 + seg * 16 + e < rows) out[e] = tile[seg * 16 + e][tc];
  }
}

struct AccumSegs {
  float* dst[MR_MAX_SEGMENTS];
  const float* src[MR_MAX_SEGMENTS];
  long long n[MR_MAX_SEGMENTS];
};

__global__ void accumulate_multi_kernel(AccumSegs segs) {
  const int s = blockIdx.y;
  float* __restrict__ d = segs.dst[s];
  const float* __restrict__ a = segs.src[s];
  const long long n = segs.n[s];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) d[i] += a[i];
}

struct ZeroSegs {
  void* dst[MR_MAX_SEGMENTS];
  long long n16[MR_MAX_SEGMENTS];   // 16-byte vectors
};
// zero fill of several 16-byte-aligned buffers in ONE launch (zero_grad: the flat gradient buffers + the scratch arena)
__global__ void zero_multi_kernel(ZeroSegs segs) {
  const int s = blockIdx.y;
  uint4* __restrict__ d = (uint4*)segs.dst[s];
  const long long n = segs.n16[s];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) d[i] = make_uint4(0, 0, 0, 0);
}

// ---------------------------------------------------------------- fused Adam over one flat buffer
// hyper (device, f32[8]): lr, beta1, beta2, eps, weight_decay, step (as float), unused, unused
// Semantics = torch.optim.Adam (no amsgrad, L2 weight decay added to the gradient).
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, const float* __restrict__ hyper) {
  // hyper[5] = number of COMPLETED steps: this launch is step hyper[5] + 1 and only READS the counter.  The launch that
  // follows it in the stream advances it -- mr_prep_batch(tick = hyper), or mr_opt_tick when there is nothing to prepare (a
  // one-thread "tick" launch in front of every update was 5 us of launch floor per step; an arrival counter inside this kernel
  // serialised 4096 returning atomics on one address: 38 -> 157 us, measured).
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], step = hyper[5] + 1.f;
  // hyper[6]: scale applied to the raw gradient (data parallel: 1 / world size folded into the update instead of a
  // separate pass over the flat gradient buffer after the all-reduce); 0 = unset = 1
  const float gs = hyper[6] != 0.f ? hyper[6] : 1.f;
  const float bc1 = 1.f - powf(b1, step);
  const float bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const long long nv = n / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nv;
       i += (long long)gridDim.x * blockDim.x) {
    f32x4 pp = ((f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gr = gg[j] * gs;
      if (wd != 0.f) gr += wd * pp[j];
      mm[j] = b1 * mm[j] + (1.f - b1) * gr;
      vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
      pp[j] -= step_size * (mm[j] / denom);
    }
    ((f32x4*)p)[i] = pp;
    ((f32x4*)m)[i] = mm;
    ((f32x4*)v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = nv * 4; i < n; ++i) {
      float gr = g[i] * gs;
      if (wd != 0.f) gr += wd * p[i];
      m[i] = b1 * m[i] + (1.f - b1) * gr;
      v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
      p[i] -= step_size * (m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps));
    }
}


// SGD with momentum (torch.optim.SGD semantics: buf = mu*buf + g(+wd*p); p -= lr*buf; first step buf = g)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n,
                           const float* __restrict__ hyper) {
  const float lr = hyper[0], mu = hyper[1], wd = hyper[4], step = hyper[5] + 1.f;   // see adam_kernel
  const float gs = hyper[6] != 0.f ? hyper[6] : 1.f;   // gradient scale (1 / world size), see adam_kernel
  const bool first = step <= 1.f;
  const long long nv = n / 4;      // the flat buffers are 256-byte aligned and padded to 64 elements (optim.py)
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pp = ((f32x4*)p)[i], bb = ((f32x4*)buf)[i];
    const f32x4 gg = ((const f32x4*)g)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * gs + wd * pp[j];
      const float b = first ? gr : mu * bb[j] + gr;
      bb[j] = b;
      pp[j] -= lr * b;
    }
    ((f32x4*)buf)[i] = bb;
    ((f32x4*)p)[i] = pp;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = nv * 4; i < n; ++i) {
      const float gr = g[i] * gs + wd * p[i];
      const float b = first ? gr : mu * buf[i] + gr;
      buf[i] = b;
      p[i] -= lr * b;
    }
}

// ---- phase timer (common.h)
static int g_phase_on = 0;
struct PhaseRec { int id; double work; hipEvent_t e0, e1; };
static std::vector<PhaseRec> g_phase_recs;
static std::mutex g_phase_mutex;
bool phase_timer_on() { return __atomic_load_n(&g_phase_on, __ATOMIC_RELAXED) != 0; }
void phase_mark(int id, bool end, double work, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_phase_mutex);
  if (!end) {
    PhaseRec r;
    r.id = id; r.work = work; r.e0 = nullptr; r.e1 = nullptr;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, stream);
    g_phase_recs.push_back(r);
  } else {
    for (size_t i = g_phase_recs.size(); i-- > 0;)
      if (g_phase_recs[i].id == id) { (void)hipEventRecord(g_phase_recs[i].e1, stream); break; }
  }
}

}  // namespace mr

using namespace mr;

#define DISPATCH_T(dtype, ...)                                   \
  if ((dtype) == MR_F32) { typedef float T; __VA_ARGS__; }       \
  else if ((dtype) == MR_BF16) { typedef bf16_t T; __VA_ARGS__; } \
  else { mr::set_error("bad dtype %d", (dtype)); return MR_ERR_DTYPE; }

extern "C" {

const char* mr_last_error(void) { return mr::g_err; }

// Phase timer: mr_phase_timer(1) starts recording (and drops earlier records), mr_phase_timer(0) stops.  mr_phase_read(id, ...)
// synchronises with the recorded events and returns the number of records of phase `id`, their total milliseconds and total
// `work`.  Not capturable: for eager measurement passes only.  Host only.
int mr_phase_timer(int on) {
  std::lock_guard<std::mutex> lock(mr::g_phase_mutex);
  const int old = mr::g_phase_on;
  if (on) {
    for (auto& r : mr::g_phase_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    mr::g_phase_recs.clear();
  }
  __atomic_store_n(&mr::g_phase_on, on ? 1 : 0, __ATOMIC_RELAXED);
  return old;
}
int mr_phase_read(int id, double* total_ms, double* total_work) {
  std::lock_guard<std::mutex> lock(mr::g_phase_mutex);
  int n = 0;
  double ms = 0.0, work = 0.0;
  for (auto& r : mr::g_phase_recs)
    if (r.id == id) {
      float t = 0.f;
      if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
        ms += t; work += r.work; ++n;
      }
    }
  if (total_ms) *total_ms = ms;
  if (total_work) *total_work = work;
  return n;
}
int mr_abi_version(void) { return MR_ABI_VERSION; }

int mr_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cpad,
                    hipStream_t stream) {
  MR_CHECK_ARG(Cpad >= C, "mr_nchw_to_nhwc: Cpad < C");
  const long long total = (long long)N * H * W * Cpad;
  DISPATCH_T(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       src, (T*)dst, N, C, H, W, Cpad));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int ld,
                    hipStream_t stream) {
  const long long total = (long long)N * H * W * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)src, dst, N, C, H, W, ld));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_cast(int src_dtype, const void* src, int dst_dtype, void* dst, long long n, hipStream_t stream) {
  const int g = grid_for(n, 256);
  if (src_dtype == MR_F32 && dst_dtype == MR_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, stream, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == MR_BF16 && dst_dtype == MR_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, stream, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == MR_F32 && dst_dtype == MR_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, stream, (const float*)src, (float*)dst, n);
  else if (src_dtype == MR_BF16 && dst_dtype == MR_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, n);
  else { mr::set_error("mr_cast: bad dtypes %d -> %d", src_dtype, dst_dtype); return MR_ERR_DTYPE; }
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_relu_bwd(int dtype, const void* dy, const void* y, void* dx, long long n, hipStream_t stream) {
  MR_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)dx & 15) == 0,
               "mr_relu_bwd: pointers must be 16-byte aligned");
  DISPATCH_T(dtype, hipLaunchKernelGGL((relu_bwd_kernel<T>), dim3(grid_for(n / VecOf<T>::N, 256)), dim3(256), 0,
                                       stream, (const T*)dy, (const T*)y, (T*)dx, n));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_add(int dtype, const void* a, const void* b, void* out, long long n, int relu, hipStream_t stream) {
  MR_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)out & 15) == 0,
               "mr_add: pointers must be 16-byte aligned");
  DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for(n / VecOf<T>::N, 256)), dim3(256), 0, stream,
                                       (const T*)a, (const T*)b, (T*)out, n, relu));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_colsum(int dtype, const void* x, float* out, int P, int C, long long ld, int perm_h, hipStream_t stream) {
  MR_CHECK_ARG(P > 0 && C > 0, "mr_colsum: bad shape");
  MR_CHECK_ARG(perm_h == 0 || C % (4 * perm_h) == 0, "mr_colsum: C must be a multiple of 4*perm_h");
  const int colg = cdiv(C, 64);
  int splits = 1024 / colg;
  if (splits < 1) splits = 1;
  if (splits > cdiv(P, 64)) splits = cdiv(P, 64);
  const int rpb = cdiv(P, splits);
  splits = cdiv(P, rpb);
  DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), dim3(colg, splits), dim3(256), 0, stream, (const T*)x,
                                       out, P, C, ld, rpb, perm_h));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_permute_021(int dtype, const void* src, void* dst, int A, int B, int C, hipStream_t stream) {
  const int vec = dtype == MR_F32 ? 4 : 8;
  MR_CHECK_ARG(C % vec == 0, "mr_permute_021: C (%d) must be a multiple of %d", C, vec);
  MR_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "mr_permute_021: 16-byte alignment");
  const long long total = (long long)A * B * (C / vec);
  DISPATCH_T(dtype, hipLaunchKernelGGL((permute_021_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       (const T*)src, (T*)dst, A, B, C));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_prep_conv_weight(int dtype, const float* src, long long sk, long long sc, long long sr, long long ss,
                        void* dst_krsc, void* dst_crsk, int K, int C, int R, int S, int Cpad, int ldk,
                        hipStream_t stream) {
  MR_CHECK_ARG(Cpad >= C && ldk >= K, "mr_prep_conv_weight: Cpad < C or ldk < K");
  MR_CHECK_ARG(dst_crsk == nullptr || Cpad == C, "mr_prep_conv_weight: crsk image requires Cpad == C");
  const long long total = (long long)K * R * S * Cpad;
  DISPATCH_T(dtype, hipLaunchKernelGGL((prep_conv_weight_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                       stream, src, sk, sc, sr, ss, (T*)dst_krsc, (T*)dst_crsk, K, C, R, S, Cpad,
                                       ldk));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_prep_matrix(int dtype, const float* src, int lds, void* dst_n, int ldn, void* dst_t, int ldt, int R, int C,
                   int perm_h, hipStream_t stream) {
  MR_CHECK_ARG(perm_h == 0 || R % (4 * perm_h) == 0, "mr_prep_matrix: R must be a multiple of 4*perm_h");
  const long long total = (long long)R * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((prep_matrix_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream,
                                       src, lds, (T*)dst_n, ldn, (T*)dst_t, ldt, R, C, perm_h));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_prep_bias(const float* a, const float* b, float* dst, int R, int perm_h, hipStream_t stream) {
  MR_CHECK_ARG(perm_h == 0 || R % (4 * perm_h) == 0, "mr_prep_bias: R must be a multiple of 4*perm_h");
  hipLaunchKernelGGL(prep_bias_kernel, dim3(cdiv(R, 256)), dim3(256), 0, stream, a, b, dst, R, perm_h);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// sizeof(mr_prep_job) as compiled into the library (host-side bindings check their struct mirror against it)
int mr_sizeof_prep_job(void) { return (int)sizeof(mr_prep_job); }

__global__ void opt_tick_kernel(float* hyper) { hyper[5] += 1.f; }

int mr_opt_tick(float* hyper, hipStream_t stream) {
  MR_CHECK_ARG(hyper != nullptr, "mr_opt_tick: null");
  hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(1), 0, stream, hyper);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_prep_batch(int dtype, const mr_prep_job* jobs_device, int njobs, long long total_blocks, float* tick,
                  hipStream_t stream) {
  if (njobs <= 0) return tick ? mr_opt_tick(tick, stream) : MR_OK;
  MR_CHECK_ARG(jobs_device != nullptr && njobs <= 1024, "mr_prep_batch: bad job table (at most 1024 jobs)");
  MR_CHECK_ARG(total_blocks > 0 && total_blocks < (1ll << 31), "mr_prep_batch: bad total_blocks");
  DISPATCH_T(dtype, hipLaunchKernelGGL((prep_batch_kernel<T>), dim3((unsigned)total_blocks), dim3(256), 0, stream,
                                       jobs_device, njobs, tick));
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_accumulate_multi(int count, float* const* dst, const float* const* src, const long long* n,
                        hipStream_t stream) {
  if (count <= 0) return MR_OK;
  MR_CHECK_ARG(count <= MR_MAX_SEGMENTS, "mr_accumulate_multi: at most %d segments per call", MR_MAX_SEGMENTS);
  AccumSegs segs;
  long long nmax = 0;
  for (int i = 0; i < count; ++i) {
    MR_CHECK_ARG(dst[i] != nullptr && src[i] != nullptr && n[i] >= 0, "mr_accumulate_multi: bad segment %d", i);
    segs.dst[i] = dst[i];
    segs.src[i] = src[i];
    segs.n[i] = n[i];
    if (n[i] > nmax) nmax = n[i];
  }
  if (nmax == 0) return MR_OK;
  hipLaunchKernelGGL(accumulate_multi_kernel, dim3(grid_for(nmax, 256, 256), count), dim3(256), 0, stream, segs);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_zero_multi(int count, void* const* dst, const long long* bytes, hipStream_t stream) {
  if (count <= 0) return MR_OK;
  MR_CHECK_ARG(count <= MR_MAX_SEGMENTS, "mr_zero_multi: at most %d segments per call", MR_MAX_SEGMENTS);
  ZeroSegs segs;
  long long nmax = 0;
  for (int i = 0; i < count; ++i) {
    MR_CHECK_ARG(dst[i] != nullptr && bytes[i] >= 0 && (((uintptr_t)dst[i]) & 15) == 0 && (bytes[i] & 15) == 0,
                 "mr_zero_multi: segment %d must be 16-byte aligned with a size that is a multiple of 16", i);
    segs.dst[i] = dst[i];
    segs.n16[i] = bytes[i] / 16;
    if (segs.n16[i] > nmax) nmax = segs.n16[i];
  }
  for (int i = count; i < MR_MAX_SEGMENTS; ++i) { segs.dst[i] = nullptr; segs.n16[i] = 0; }
  if (nmax == 0) return MR_OK;
  hipLaunchKernelGGL(zero_multi_kernel, dim3(grid_for(nmax, 256, 2048), count), dim3(256), 0, stream, segs);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_adam_step(float* p, const float* g, float* m, float* v, long long n, float* hyper, hipStream_t stream) {
  MR_CHECK_ARG(((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)m & 15) == 0 &&
                   ((uintptr_t)v & 15) == 0,
               "mr_adam_step: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, stream, p, g, m, v, n, (const float*)hyper);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

int mr_sgd_step(float* p, const float* g, float* buf, long long n, float* hyper, hipStream_t stream) {
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, stream, p, g, buf, n, (const float*)hyper);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
