// Deformable position-sensitive RoI pooling for gfx950.
// Replaces assets/ops/dcn/src/deform_pool_cuda_kernel.cu:52-143 (forward) and :146-268 (backward) behind the entry points
// of deform_pool_cuda.cpp:30-77.  Exported by the reference's assets.ops.dcn package, used by no model or YAML
// (SURVEY.md §8 f4).
//
// Tensors keep the reference extension's contract: data [B][C][H][W] f32 contiguous, rois [R][5] = (batch index,
// x1, y1, x2, y2), trans [R][2*classes][part][part], out / top_count [R][output_dim][P][P], all caller-allocated.
// An output element averages up to sample_per_part^2 bilinear samples of ONE channel plane,
// c = (ctop*G + gh)*G + gw, taken on a regular sub-grid of its bin; the bin is shifted by trans * roi size.
//
// HBM-bound gather / scatter on small tensors.  Differences from the reference by design:
//  * forward: one lane per output element in output order, so a wave's stores are contiguous and its gathers walk
//    neighbouring x positions of one plane;
//  * backward: lanes run over ctop FASTEST (thread <-> (n, ph, pw, ctop)): the lanes of a wave then scatter into
//    different channel planes (no same-address atomics inside a wave) and share one (n, class, part_h, part_w)
//    offset cell, so the two offset gradients are accumulated per lane over the samples, reduced across the wave
//    with DPP/shuffle adds and leave as ONE atomic pair per wave -- the reference issues two same-address atomics per
//    SAMPLE per thread (2 * spp^2 * 64 per wave).
#include "common.h"
#include "../../include/megreader_hip.h"

#pragma clang fp contract(off)   // the bin arithmetic must round like the reference's float expressions

namespace mr {

struct PsRoiGeom {
  int B, C, H, W, R, no_trans, output_dim, group_size, pooled, part_size, spp, num_classes, ch_per_class;
  float spatial_scale, trans_std;
};

struct PsRoiBin {
  float wstart, hstart, sub_w, sub_h, roi_w, roi_h;
  int batch, c, tx_idx;   // tx_idx: index of the x offset in trans (y offset at tx_idx + part*part)
};

__device__ __forceinline__ PsRoiBin psroi_bin(const PsRoiGeom& g, const float* __restrict__ rois,
                                              const float* __restrict__ trans, int n, int ctop, int ph, int pw) {
  PsRoiBin b;
  const float* r = rois + (long long)n * 5;
  b.batch = (int)r[0];
  const float x1 = roundf(r[1]) * g.spatial_scale - 0.5f;
  const float y1 = roundf(r[2]) * g.spatial_scale - 0.5f;
  const float x2 = (roundf(r[3]) + 1.f) * g.spatial_scale - 0.5f;
  const float y2 = (roundf(r[4]) + 1.f) * g.spatial_scale - 0.5f;
  b.roi_w = fmaxf(x2 - x1, 0.1f);
  b.roi_h = fmaxf(y2 - y1, 0.1f);
  const float bin_h = b.roi_h / (float)g.pooled, bin_w = b.roi_w / (float)g.pooled;
  b.sub_h = bin_h / (float)g.spp;
  b.sub_w = bin_w / (float)g.spp;
  const int part_h = (int)floorf((float)ph / (float)g.pooled * (float)g.part_size);
  const int part_w = (int)floorf((float)pw / (float)g.pooled * (float)g.part_size);
  const int class_id = ctop / g.ch_per_class;
  b.tx_idx = (((n * g.num_classes + class_id) * 2) * g.part_size + part_h) * g.part_size + part_w;
  float tx = 0.f, ty = 0.f;
  if (!g.no_trans) {
    tx = trans[b.tx_idx] * g.trans_std;
    ty = trans[b.tx_idx + g.part_size * g.part_size] * g.trans_std;
  }
  b.wstart = (float)pw * bin_w + x1;
  b.wstart += tx * b.roi_w;
  b.hstart = (float)ph * bin_h + y1;
  b.hstart += ty * b.roi_h;
  int gw = (int)floorf((float)pw * (float)g.group_size / (float)g.pooled);
  int gh = (int)floorf((float)ph * (float)g.group_size / (float)g.pooled);
  gw = min(max(gw, 0), g.group_size - 1);
  gh = min(max(gh, 0), g.group_size - 1);
  b.c = (ctop * g.group_size + gh) * g.group_size + gw;
  return b;
}

// sample position -> clamped coordinates; false when the reference skips the sample
__device__ __forceinline__ bool psroi_sample(const PsRoiGeom& g, const PsRoiBin& b, int ih, int iw, float& w,
                                             float& h) {
  w = b.wstart + (float)iw * b.sub_w;
  h = b.hstart + (float)ih * b.sub_h;
  if (w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f) return false;
  w = fminf(fmaxf(w, 0.f), (float)g.W - 1.f);
  h = fminf(fmaxf(h, 0.f), (float)g.H - 1.f);
  return true;
}

__global__ __launch_bounds__(256) void psroi_fwd_kernel(const float* __restrict__ data,
                                                         const float* __restrict__ rois,
                                                         const float* __restrict__ trans, float* __restrict__ top,
                                                         float* __restrict__ top_count, PsRoiGeom g) {
  const long long total = (long long)g.R * g.output_dim * g.pooled * g.pooled;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % g.pooled);
    const int ph = (int)((idx / g.pooled) % g.pooled);
    const int ctop = (int)((idx / g.pooled / g.pooled) % g.output_dim);
    const int n = (int)(idx / g.pooled / g.pooled / g.output_dim);
    const PsRoiBin b = psroi_bin(g, rois, trans, n, ctop, ph, pw);
    const float* plane = data + ((long long)b.batch * g.C + b.c) * g.H * g.W;
    float sum = 0.f;
    int cnt = 0;
    for (int ih = 0; ih < g.spp; ++ih)
      for (int iw = 0; iw < g.spp; ++iw) {
        float w, h;
        if (!psroi_sample(g, b, ih, iw, w, h)) continue;
        const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        const float dx = w - (float)x0, dy = h - (float)y0;
        const float v00 = plane[y0 * g.W + x0], v01 = plane[y1 * g.W + x0];
        const float v10 = plane[y0 * g.W + x1], v11 = plane[y1 * g.W + x1];
        sum += (1.f - dx) * (1.f - dy) * v00 + (1.f - dx) * dy * v01 + dx * (1.f - dy) * v10 + dx * dy * v11;
        ++cnt;
      }
    top[idx] = cnt == 0 ? 0.f : sum / (float)cnt;
    top_count[idx] = (float)cnt;
  }
}

// thread <-> (n, ph, pw, ctop), ctop fastest; grid-stride in whole waves so that a wave's lanes always share (n, ph, pw)
__global__ __launch_bounds__(256) void psroi_bwd_kernel(const float* __restrict__ top_diff,
                                                         const float* __restrict__ top_count,
                                                         const float* __restrict__ data,
                                                         const float* __restrict__ rois,
                                                         const float* __restrict__ trans, float* __restrict__ data_diff,
                                                         float* __restrict__ trans_diff, PsRoiGeom g) {
  const int lane = threadIdx.x & 63;
  const int ct_waves = (g.output_dim + 63) / 64;
  const long long cells = (long long)g.R * g.pooled * g.pooled;
  const long long total_waves = cells * ct_waves;
  const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long wave_stride = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long wv = wave0; wv < total_waves; wv += wave_stride) {
    const long long cell = wv / ct_waves;
    const int ctop = (int)(wv % ct_waves) * 64 + lane;
    const int pw = (int)(cell % g.pooled);
    const int ph = (int)((cell / g.pooled) % g.pooled);
    const int n = (int)(cell / g.pooled / g.pooled);
    const bool active = ctop < g.output_dim;
    float gx = 0.f, gy = 0.f;
    int key = -1;
    if (active) {
      const long long idx = (((long long)n * g.output_dim + ctop) * g.pooled + ph) * g.pooled + pw;
      const float cnt = top_count[idx];
      const PsRoiBin b = psroi_bin(g, rois, trans, n, ctop, ph, pw);
      key = b.tx_idx;
      if (cnt > 0.f) {
        const float dv = top_diff[idx] / cnt;
        const long long base = ((long long)b.batch * g.C + b.c) * g.H * g.W;
        const float* plane = data + base;
        float* dplane = data_diff + base;
        for (int ih = 0; ih < g.spp; ++ih)
          for (int iw = 0; iw < g.spp; ++iw) {
            float w, h;
            if (!psroi_sample(g, b, ih, iw, w, h)) continue;
            const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
            const float dx = w - (float)x0, dy = h - (float)y0;
            atomicAdd(dplane + y0 * g.W + x0, (1.f - dx) * (1.f - dy) * dv);
            atomicAdd(dplane + y1 * g.W + x0, (1.f - dx) * dy * dv);
            atomicAdd(dplane + y0 * g.W + x1, dx * (1.f - dy) * dv);
            atomicAdd(dplane + y1 * g.W + x1, dx * dy * dv);
            if (g.no_trans) continue;
            const float u00 = plane[y0 * g.W + x0], u01 = plane[y1 * g.W + x0];
            const float u10 = plane[y0 * g.W + x1], u11 = plane[y1 * g.W + x1];
            float ddx = (u11 * dy + u10 * (1.f - dy) - u01 * dy - u00 * (1.f - dy)) * g.trans_std * dv;
            ddx *= b.roi_w;
            float ddy = (u11 * dx + u01 * (1.f - dx) - u10 * dx - u00 * (1.f - dx)) * g.trans_std * dv;
            ddy *= b.roi_h;
            gx += ddx;
            gy += ddy;
          }
      }
    }
    if (g.no_trans) continue;
    // lanes of one class share the offset cell: one atomic pair per (wave, class) instead of per sample
    const int pp = g.part_size * g.part_size;
    unsigned long long todo = __ballot(active);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int k = __shfl(key, leader, 64);
      const bool mine = active && key == k;
      const float sx = wave_sum(mine ? gx : 0.f), sy = wave_sum(mine ? gy : 0.f);
      if (lane == leader) {
        atomicAdd(trans_diff + k, sx);
        atomicAdd(trans_diff + k + pp, sy);
      }
      todo &= ~__ballot(mine);
    }
  }
}

static inline int grid_for(long long n, int block, int max_blocks = 8192) {
  const long long b = (n + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > max_blocks ? max_blocks : b));
}

static int psroi_geom(PsRoiGeom& g, int B, int C, int H, int W, int R, int channels_trans, int no_trans,
                      float spatial_scale, int output_dim, int group_size, int pooled_size, int part_size,
                      int sample_per_part, float trans_std, const char* who) {
  MR_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && R >= 0, "%s: bad shape", who);
  MR_CHECK_ARG(output_dim > 0 && group_size > 0 && pooled_size > 0 && part_size > 0 && sample_per_part > 0,
               "%s: bad pooling parameters", who);
  MR_CHECK_ARG(output_dim * group_size * group_size <= C, "%s: output_dim*group_size^2 (%d) exceeds the %d input channels",
               who, output_dim * group_size * group_size, C);
  g.num_classes = no_trans ? 1 : channels_trans / 2;
  MR_CHECK_ARG(g.num_classes > 0 && output_dim % g.num_classes == 0 || no_trans,
               "%s: output_dim (%d) must be a multiple of the %d offset classes", who, output_dim, g.num_classes);
  g.ch_per_class = no_trans ? output_dim : output_dim / g.num_classes;
  g.B = B; g.C = C; g.H = H; g.W = W; g.R = R; g.no_trans = no_trans ? 1 : 0; g.output_dim = output_dim;
  g.group_size = group_size; g.pooled = pooled_size; g.part_size = part_size; g.spp = sample_per_part;
  g.spatial_scale = spatial_scale; g.trans_std = trans_std;
  return MR_OK;
}

}  // namespace mr

using namespace mr;

extern "C" {

// deform_pool_cuda.cpp:30-52 (deform_psroi_pooling_cuda_forward): writes out and top_count, [R][output_dim][P][P] f32
int mr_deform_psroi_fwd(const float* data, const float* rois, const float* trans, float* out, float* top_count, int B,
                        int C, int H, int W, int R, int channels_trans, int no_trans, float spatial_scale,
                        int output_dim, int group_size, int pooled_size, int part_size, int sample_per_part,
                        float trans_std, hipStream_t stream) {
  PsRoiGeom g;
  const int rc = psroi_geom(g, B, C, H, W, R, channels_trans, no_trans, spatial_scale, output_dim, group_size,
                            pooled_size, part_size, sample_per_part, trans_std, "mr_deform_psroi_fwd");
  if (rc != MR_OK) return rc;
  const long long total = (long long)R * output_dim * pooled_size * pooled_size;
  if (total == 0) return MR_OK;
  hipLaunchKernelGGL(psroi_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, data, rois, trans, out,
                     top_count, g);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

// deform_pool_cuda.cpp:54-77 (deform_psroi_pooling_cuda_backward): ACCUMULATES into data_diff [B][C][H][W] and
// trans_diff (same shape as trans; may be null when no_trans) -- the caller zero-fills them, as the reference's
// functions/deform_pool.py:57-59 does.
int mr_deform_psroi_bwd(const float* out_grad, const float* data, const float* rois, const float* trans,
                        const float* top_count, float* data_diff, float* trans_diff, int B, int C, int H, int W, int R,
                        int channels_trans, int no_trans, float spatial_scale, int output_dim, int group_size,
                        int pooled_size, int part_size, int sample_per_part, float trans_std, hipStream_t stream) {
  PsRoiGeom g;
  const int rc = psroi_geom(g, B, C, H, W, R, channels_trans, no_trans, spatial_scale, output_dim, group_size,
                            pooled_size, part_size, sample_per_part, trans_std, "mr_deform_psroi_bwd");
  if (rc != MR_OK) return rc;
  MR_CHECK_ARG(no_trans || trans_diff, "mr_deform_psroi_bwd: trans_diff is null");
  const long long waves = (long long)R * pooled_size * pooled_size * ((output_dim + 63) / 64);
  if (waves == 0) return MR_OK;
  hipLaunchKernelGGL(psroi_bwd_kernel, dim3(grid_for(waves * 64, 256)), dim3(256), 0, stream, out_grad, top_count, data,
                     rois, trans, data_diff, trans_diff, g);
  MR_CHECK_LAUNCH();
  return MR_OK;
}

}  // extern "C"
