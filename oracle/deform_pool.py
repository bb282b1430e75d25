"""Oracle restatement (numpy float64, explicit loops) of the reference's CUDA-only deformable PS-RoI pooling.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows assets/ops/dcn/src/deform_pool_cuda_kernel.cu line by line in meaning:
    :32-51    bilinear_interp (floor / ceil corners, weights from the fractional parts)
    :52-143   DeformablePSROIPoolForwardKernel: RoI corners rounded and scaled (:83-86), width / height floored at 0.1
              (:89-90), bin and sub-bin sizes (:93-97), part cell and class of the offset (:99-103), bin start shifted by
              trans * trans_std * roi size (:105-108), group cell -> input channel (:112-127), samples outside
              [-0.5, W-0.5] x [-0.5, H-0.5] skipped, the rest clamped (:122-127), mean over the counted samples (:132-133)
    :146-268  DeformablePSROIPoolBackwardAccKernel: diff / count scattered with the bilinear weights (:235-238), offset
              gradient from the corner values (:248-255)
and the host wrappers deform_pool_cuda.cpp:30-77 (num_classes = channels_trans / 2, channels_each_class).

PARITY STATUS: pinned (round 4) against the reference's own `deform_pool_cuda` module, compiled for gfx950 from
assets/ops/dcn/src/deform_pool_cuda.cpp + deform_pool_cuda_kernel.cu where they lie (oracle/build_ref_ext.sh) and run on the
MI355X by oracle/gen_golden_dcn.py: six cases in tests/golden/dcn_reference_ext.npz (with / without offsets, several classes,
group sizes, 70 output channels, a single 1 x 1 bin), checked by tests/test_oracle_dcn_pinned_cpu.py -- sample counts exactly,
outputs to 2e-6, gradients to 1e-5 (the reference is float32 with float atomics).  Older anchors
(tests/test_oracle_deform_pool.py): no_trans with group_size = 1 on an aligned RoI equals average pooling of the bin's samples
computed independently; the analytic offset gradient equals central finite differences of the forward; the data gradient is
the exact adjoint of the (linear in data) forward.
"""
import math

import numpy as np


def _geom(rois, trans, n, ctop, ph, pw, p):
    r = rois[n]
    x1 = round_half_away(r[1]) * p['spatial_scale'] - 0.5
    y1 = round_half_away(r[2]) * p['spatial_scale'] - 0.5
    x2 = (round_half_away(r[3]) + 1.0) * p['spatial_scale'] - 0.5
    y2 = (round_half_away(r[4]) + 1.0) * p['spatial_scale'] - 0.5
    roi_w, roi_h = max(x2 - x1, 0.1), max(y2 - y1, 0.1)
    P, spp, part, G = p['pooled'], p['spp'], p['part_size'], p['group_size']
    bin_h, bin_w = roi_h / P, roi_w / P
    sub_h, sub_w = bin_h / spp, bin_w / spp
    part_h, part_w = int(math.floor(ph / P * part)), int(math.floor(pw / P * part))
    class_id = ctop // p['ch_per_class']
    tx = ty = 0.0
    if not p['no_trans']:
        tx = trans[n, class_id * 2, part_h, part_w] * p['trans_std']
        ty = trans[n, class_id * 2 + 1, part_h, part_w] * p['trans_std']
    wstart = pw * bin_w + x1 + tx * roi_w
    hstart = ph * bin_h + y1 + ty * roi_h
    gw = min(max(int(math.floor(pw * G / P)), 0), G - 1)
    gh = min(max(int(math.floor(ph * G / P)), 0), G - 1)
    c = (ctop * G + gh) * G + gw
    return int(r[0]), c, wstart, hstart, sub_w, sub_h, roi_w, roi_h, class_id, part_h, part_w


def round_half_away(v):
    """C `round()`: halves away from zero (numpy / python round to even)."""
    return math.floor(abs(v) + 0.5) * (1.0 if v >= 0 else -1.0)


def _params(data, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part,
            trans_std):
    num_classes = 1 if no_trans else trans.shape[1] // 2
    return dict(no_trans=bool(no_trans), spatial_scale=float(spatial_scale), output_dim=output_dim,
                group_size=group_size, pooled=pooled_size, part_size=part_size, spp=sample_per_part,
                trans_std=float(trans_std), ch_per_class=output_dim if no_trans else output_dim // num_classes)


def _samples(p, H, W, wstart, hstart, sub_w, sub_h):
    for ih in range(p['spp']):
        for iw in range(p['spp']):
            w, h = wstart + iw * sub_w, hstart + ih * sub_h
            if w < -0.5 or w > W - 0.5 or h < -0.5 or h > H - 0.5:
                continue
            w, h = min(max(w, 0.0), W - 1.0), min(max(h, 0.0), H - 1.0)
            x0, x1, y0, y1 = int(math.floor(w)), int(math.ceil(w)), int(math.floor(h)), int(math.ceil(h))
            yield x0, x1, y0, y1, w - x0, h - y0


def psroi_forward(data, rois, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
                  sample_per_part, trans_std):
    """data [B,C,H,W], rois [R,5], trans [R,2*classes,part,part] -> (out, count) [R,output_dim,P,P] float64."""
    data, rois = np.asarray(data, np.float64), np.asarray(rois, np.float64)
    trans = None if no_trans else np.asarray(trans, np.float64)
    p = _params(data, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part,
                trans_std)
    _, _, H, W = data.shape
    R, P = rois.shape[0], pooled_size
    out, cnt = np.zeros((R, output_dim, P, P)), np.zeros((R, output_dim, P, P))
    for n in range(R):
        for ctop in range(output_dim):
            for ph in range(P):
                for pw in range(P):
                    b, c, ws, hs, sw, sh, _, _, _, _, _ = _geom(rois, trans, n, ctop, ph, pw, p)
                    plane, s, k = data[b, c], 0.0, 0
                    for x0, x1, y0, y1, dx, dy in _samples(p, H, W, ws, hs, sw, sh):
                        s += ((1 - dx) * (1 - dy) * plane[y0, x0] + (1 - dx) * dy * plane[y1, x0] +
                              dx * (1 - dy) * plane[y0, x1] + dx * dy * plane[y1, x1])
                        k += 1
                    out[n, ctop, ph, pw] = 0.0 if k == 0 else s / k
                    cnt[n, ctop, ph, pw] = k
    return out, cnt


def psroi_backward(out_grad, data, rois, trans, count, no_trans, spatial_scale, output_dim, group_size, pooled_size,
                   part_size, sample_per_part, trans_std):
    """-> (data_grad [B,C,H,W], trans_grad like trans or None) float64."""
    out_grad, data, rois = (np.asarray(a, np.float64) for a in (out_grad, data, rois))
    trans = None if no_trans else np.asarray(trans, np.float64)
    p = _params(data, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part,
                trans_std)
    _, _, H, W = data.shape
    R, P = rois.shape[0], pooled_size
    dgrad = np.zeros_like(data)
    tgrad = None if no_trans else np.zeros_like(trans)
    for n in range(R):
        for ctop in range(output_dim):
            for ph in range(P):
                for pw in range(P):
                    if count[n, ctop, ph, pw] <= 0:
                        continue
                    dv = out_grad[n, ctop, ph, pw] / count[n, ctop, ph, pw]
                    b, c, ws, hs, sw, sh, roi_w, roi_h, cls, part_h, part_w = _geom(rois, trans, n, ctop, ph, pw, p)
                    plane = data[b, c]
                    for x0, x1, y0, y1, dx, dy in _samples(p, H, W, ws, hs, sw, sh):
                        dgrad[b, c, y0, x0] += (1 - dx) * (1 - dy) * dv
                        dgrad[b, c, y1, x0] += (1 - dx) * dy * dv
                        dgrad[b, c, y0, x1] += dx * (1 - dy) * dv
                        dgrad[b, c, y1, x1] += dx * dy * dv
                        if no_trans:
                            continue
                        u00, u01, u10, u11 = plane[y0, x0], plane[y1, x0], plane[y0, x1], plane[y1, x1]
                        gx = (u11 * dy + u10 * (1 - dy) - u01 * dy - u00 * (1 - dy)) * p['trans_std'] * dv * roi_w
                        gy = (u11 * dx + u01 * (1 - dx) - u10 * dx - u00 * (1 - dx)) * p['trans_std'] * dv * roi_h
                        tgrad[n, cls * 2, part_h, part_w] += gx
                        tgrad[n, cls * 2 + 1, part_h, part_w] += gy
    return dgrad, tgrad


def random_case(seed, B=2, C=8, H=12, W=14, R=5, output_dim=2, group_size=2, pooled=3, part=3, spp=2, classes=1,
                no_trans=False):
    """Seeded inputs shared by the CPU and GPU tests: RoIs partly outside the map, non-integer corners, halves."""
    g = np.random.default_rng(seed)
    data = g.standard_normal((B, C, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = g.integers(0, B, R)
    x1, y1 = g.uniform(-3, W * 2 - 4, R), g.uniform(-3, H * 2 - 4, R)
    rois[:, 1], rois[:, 2] = x1, y1
    rois[:, 3], rois[:, 4] = x1 + g.uniform(0, W * 1.5, R), y1 + g.uniform(0, H * 1.5, R)
    rois[0, 1:] = [2.5, 3.5, 9.5, 10.5]                      # halves: C round() goes away from zero
    trans = (g.uniform(-1, 1, (R, 2 * classes, part, part))).astype(np.float32)
    kw = dict(no_trans=no_trans, spatial_scale=0.5, output_dim=output_dim, group_size=group_size, pooled_size=pooled,
              part_size=part, sample_per_part=spp, trans_std=0.0 if no_trans else 0.2)
    return data, rois, trans, kw
