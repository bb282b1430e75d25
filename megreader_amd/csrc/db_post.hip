// DB (differentiable binarization) detector post-processing: the per-pixel stages on the GPU.
// Replaces the cv2 calls of structure/representers/seg_detector_representer.py:63-168 (`boxes_from_bitmap`):
//   cv2.findContours(bitmap)           -> connected components (8-connectivity) by union-find, then the RUN END POINTS of
//                                         every component (the only pixels that can be convex-hull vertices), compacted
//                                         into a list the host turns into min-area rectangles (cv2.minAreaRect);
//   box_score_fast (fillPoly + mean)   -> mean of the probability map over the pixels inside each candidate box.
// The geometry on a few hundred hull points per image (hull, rotating calipers, unclip, ordering) stays on the host
// (megreader_amd/structure/db_geometry.py): it is O(components), not O(pixels).
// HBM-bound on one H x W map per image (640 x 640 = 1.6 MB f32): a few microseconds of traffic, launch-latency-bound.
#include "common.h"
#include "../../include/megreader_hip.h"

namespace mr {

__device__ __forceinline__ int cc_find(const int* __restrict__ L, int a) {
  int p = L[a];
  while (p != a) {
    a = p;
    p = L[a];
  }
  return a;
}

// lock-free union by smaller root index (Playne & Hawick style): the root of a component ends up its raster-first pixel
__device__ __forceinline__ void cc_union(int* __restrict__ L, int a, int b) {
  while (true) {
    a = cc_find(L, a);
    b = cc_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&L[a], b);      // hang the larger root under the smaller one
    if (old == a) return;
    a = old;                                   // somebody re-rooted a meanwhile: retry from there
  }
}

// labels[n][p] = p (image-local pixel index) where prob > thresh, else -1
__global__ void db_cc_init_kernel(const float* __restrict__ prob, float thresh, int* __restrict__ labels,
                                  long long total, int hw) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x)
    labels[i] = prob[i] > thresh ? (int)(i % hw) : -1;
}

// union every foreground pixel with its already-visited 8-neighbours (W, NW, N, NE)
__global__ void db_cc_merge_kernel(int* __restrict__ labels, int N, int H, int W) {
  const long long total = (long long)N * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int hw = H * W;
    const int n = (int)(i / hw), p = (int)(i - (long long)n * hw);
    int* L = labels + (long long)n * hw;
    if (L[p] < 0) continue;
    const int y = p / W, x = p - y * W;
    if (x > 0 && L[p - 1] >= 0) cc_union(L, p, p - 1);
    if (y > 0) {
      if (L[p - W] >= 0) cc_union(L, p, p - W);
      if (x > 0 && L[p - W - 1] >= 0) cc_union(L, p, p - W - 1);
      if (x < W - 1 && L[p - W + 1] >= 0) cc_union(L, p, p - W + 1);
    }
  }
}

// flatten + emit the run end points: pixel p is listed when its left or right neighbour is not foreground.
// points[k] = (image n, root, x, y); *count is advanced atomically (wave-aggregated); entries beyond `cap` are dropped
// (the count still says how many there were).
__global__ void db_cc_points_kernel(int* __restrict__ labels, int N, int H, int W, int4* __restrict__ points,
                                    int* __restrict__ count, int cap) {
  const long long total = (long long)N * H * W;
  const long long rounds = (total + (long long)gridDim.x * blockDim.x - 1) / ((long long)gridDim.x * blockDim.x);
  for (long long r = 0; r < rounds; ++r) {
    const long long i = r * (long long)gridDim.x * blockDim.x + blockIdx.x * (long long)blockDim.x + threadIdx.x;
    bool emit = false;
    int n = 0, x = 0, y = 0, root = -1;
    if (i < total) {
      const int hw = H * W;
      n = (int)(i / hw);
      const int p = (int)(i - (long long)n * hw);
      int* L = labels + (long long)n * hw;
      if (L[p] >= 0) {
        root = cc_find(L, p);
        y = p / W;
        x = p - y * W;
        emit = x == 0 || x == W - 1 || L[p - 1] < 0 || L[p + 1] < 0;
      }
    }
    // all neighbours' labels were read before anybody writes the flattened one? not needed: cc_find tolerates both
    const unsigned long long m = __ballot(emit);
    if (m) {
      const int lane = threadIdx.x & 63;
      int base = 0;
      if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, __popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1, 64);
      if (emit) {
        const int k = base + __popcll(m & ((1ull << lane) - 1ull));
        if (k < cap) points[k] = make_int4(n, root, x, y);
      }
    }
    if (i < total && root >= 0) labels[i] = root;   // flattened label (a root keeps itself; others point at a root)
  }
}

// mean of prob inside (or on the border of) each convex quadrilateral: one workgroup per box.
// boxes: [B][9] floats = image index, then 4 vertices (x, y) in order around the quad.  out: [B][2] = sum, count.
__global__ __launch_bounds__(256) void db_box_score_kernel(const float* __restrict__ prob, int H, int W,
                                                            const float* __restrict__ boxes, float* __restrict__ out) {
  const float* b = boxes + (long long)blockIdx.x * 9;
  const int n = (int)b[0];
  float vx[4], vy[4];
  float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    vx[k] = b[1 + 2 * k];
    vy[k] = b[2 + 2 * k];
    xmin = fminf(xmin, vx[k]); xmax = fmaxf(xmax, vx[k]);
    ymin = fminf(ymin, vy[k]); ymax = fmaxf(ymax, vy[k]);
  }
  const int x0 = max(0, (int)floorf(xmin)), x1 = min(W - 1, (int)ceilf(xmax));
  const int y0 = max(0, (int)floorf(ymin)), y1 = min(H - 1, (int)ceilf(ymax));
  // orientation of the vertex order (sign of twice the area)
  float area2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) area2 += vx[k] * vy[(k + 1) & 3] - vx[(k + 1) & 3] * vy[k];
  const float sgn = area2 >= 0.f ? 1.f : -1.f;
  const float* pm = prob + (long long)n * H * W;
  float s = 0.f, c = 0.f;
  const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
  for (int t = threadIdx.x; t < bw * bh; t += 256) {
    const int y = y0 + t / bw, x = x0 + t % bw;
    bool in = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float ex = vx[(k + 1) & 3] - vx[k], ey = vy[(k + 1) & 3] - vy[k];
      in &= sgn * (ex * ((float)y - vy[k]) - ey * ((float)x - vx[k])) >= 0.f;
    }
    if (in) {
      s += pm[(long long)y * W + x];
      c += 1.f;
    }
  }
  __shared__ float rs[4], rc[4];
  s = wave_sum(s);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = rs[0] + rs[1] + rs[2] + rs[3];
    out[2 * blockIdx.x + 1] = rc[0] + rc[1] + rc[2] + rc[3];
  }
}

static inline int grid_for(long long n, int block, int max_blocks = 4096) {
  const long long b = (n + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > max_blocks ? max_blocks : b));
}

}  // namespace mr

using namespace mr;

extern "C" {

// prob f32 [N][H][W] -> labels i32 [N][H][W] (root pixel index of the 8-connected component, -1 = background),
// points int4 [cap] = (n, root, x, y) of every run end point, *count = how many (may exceed cap: enlarge and re-run).
// *count must be zero on entry.
int mr_db_components(const float* prob, float thresh, int* labels, void* points, int* count, int cap, int N, int H, int W,
                     hipStream_t stream) {
  MR_CHECK_ARG(N > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 31) && cap >= 0, "mr_db_components: bad shape");
  const long long total = (long long)N * H * W;
  hipLaunchKernelGGL(db_cc_init_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, prob, thresh, labels, total,
                     H * W);
  hipLaunchKernelGGL(db_cc_merge_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, labels, N, H, W);
  hipLaunchKernelGGL(db_cc_points_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, labels, N, H, W,
                     (int4*)points, count, cap);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// boxes f32 [B][9] (image index, 4 vertices in order), out f32 [B][2] = (sum of prob, pixel count) inside each box
int mr_db_box_scores(const float* prob, const float* boxes, float* out, int B, int N, int H, int W, hipStream_t stream) {
  MR_CHECK_ARG(B >= 0 && N > 0 && H > 0 && W > 0, "mr_db_box_scores: bad shape");
  if (B == 0) return MR_OK;
  hipLaunchKernelGGL(db_box_score_kernel, dim3(B), dim3(256), 0, stream, prob, H, W, boxes, out);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
