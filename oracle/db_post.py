"""Oracle restatement (numpy / pure Python) of the DB detector's post-processing, `boxes_from_bitmap` of
structure/representers/seg_detector_representer.py:63-118 with its helpers (:120-168).  TEST INFRASTRUCTURE ONLY.

PARITY STATUS: "parity unpinned".  The reference calls cv2.findContours / minAreaRect / boxPoints / fillPoly / mean and
pyclipper / shapely; none of them is installed in the build image, so the reference function cannot be executed here and
no golden vector exists.  This file restates what those calls compute, stage by stage, independently of the product code
(breadth-first flood fill instead of union-find, all component pixels instead of run end points, brute-force rectangle
search over hull edges, a pixel loop for the score):
  * contours -> 8-connected components of (map > thresh) (cv2.findContours follows 8-connected foreground borders;
    RETR_LIST would also list hole borders -- not restated, see megreader_amd/structure/seg_detector_representer.py);
  * cv2.minAreaRect(contour) + boxPoints -> minimum-area enclosing rectangle of the component's pixel coordinates;
  * get_mini_boxes (:128-148) corner order; box_score_fast (:160-168): mean of pred over the pixels of the polygon with
    integer-truncated vertices (fillPoly covers interior and border);
  * unclip (:120-126): offset by area * 1.5 / perimeter with round joins, then minAreaRect again = the rectangle grown by
    that distance on every side (closed form for rectangles);
  * scaling / rounding / clipping (:110-114) with numpy's round-half-even.
"""
import math
from collections import deque

import numpy as np


def components(mask):
    """8-connected components of a bool [H, W] array, in raster order of their first pixel: list of [(x, y), ...]."""
    H, W = mask.shape
    seen = np.zeros_like(mask, dtype=bool)
    out = []
    for y in range(H):
        for x in range(W):
            if not mask[y, x] or seen[y, x]:
                continue
            q, comp = deque([(x, y)]), []
            seen[y, x] = True
            while q:
                cx, cy = q.popleft()
                comp.append((cx, cy))
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        nx, ny = cx + dx, cy + dy
                        if 0 <= nx < W and 0 <= ny < H and mask[ny, nx] and not seen[ny, nx]:
                            seen[ny, nx] = True
                            q.append((nx, ny))
            out.append(comp)
    return out


def _hull(pts):
    pts = sorted(set((p[0], p[1]) for p in pts))
    if len(pts) <= 2:
        return pts

    def half(seq):
        h = []
        for p in seq:
            while len(h) >= 2 and ((h[-1][0] - h[-2][0]) * (p[1] - h[-2][1]) -
                                   (h[-1][1] - h[-2][1]) * (p[0] - h[-2][0])) <= 0:
                h.pop()
            h.append(p)
        return h[:-1]
    return half(pts) + half(pts[::-1])


def min_rect(pts):
    """Brute force over hull edge directions: (corners in order, (side, side))."""
    h = [(float(x), float(y)) for x, y in _hull(pts)]
    if len(h) == 1:
        return [list(h[0])] * 4, (0.0, 0.0)
    if len(h) == 2:
        return [list(h[0]), list(h[1]), list(h[1]), list(h[0])], (math.dist(h[0], h[1]), 0.0)
    best = None
    for i in range(len(h)):
        (x0, y0), (x1, y1) = h[i], h[(i + 1) % len(h)]
        ln = math.hypot(x1 - x0, y1 - y0)
        ux, uy = (x1 - x0) / ln, (y1 - y0) / ln
        us = [(px - x0) * ux + (py - y0) * uy for px, py in h]
        vs = [-(px - x0) * uy + (py - y0) * ux for px, py in h]
        area = (max(us) - min(us)) * (max(vs) - min(vs))
        if best is None or area < best[0] - 1e-12:
            best = (area, x0, y0, ux, uy, min(us), max(us), min(vs), max(vs))
    _, x0, y0, ux, uy, a0, a1, b0, b1 = best
    return ([[x0 + a * ux - b * uy, y0 + a * uy + b * ux] for a, b in ((a0, b0), (a1, b0), (a1, b1), (a0, b1))],
            (a1 - a0, b1 - b0))


def mini_box(pts):
    corners, sides = min_rect(pts)
    p = sorted(corners, key=lambda c: c[0])
    i1, i4 = (0, 1) if p[1][1] > p[0][1] else (1, 0)
    i2, i3 = (2, 3) if p[3][1] > p[2][1] else (3, 2)
    return [p[i1], p[i2], p[i3], p[i4]], min(sides)


def box_score(pred, box):
    """Mean of pred over the pixels inside or on the border of the quadrilateral with truncated vertices."""
    H, W = pred.shape
    v = [(float(int(x)), float(int(y))) for x, y in box]
    xs, ys = [p[0] for p in v], [p[1] for p in v]
    x0, x1 = max(0, int(math.floor(min(xs)))), min(W - 1, int(math.ceil(max(xs))))
    y0, y1 = max(0, int(math.floor(min(ys)))), min(H - 1, int(math.ceil(max(ys))))
    area2 = sum(v[k][0] * v[(k + 1) % 4][1] - v[(k + 1) % 4][0] * v[k][1] for k in range(4))
    sgn = 1.0 if area2 >= 0 else -1.0
    s, c = 0.0, 0
    for y in range(y0, y1 + 1):
        for x in range(x0, x1 + 1):
            if all(sgn * ((v[(k + 1) % 4][0] - v[k][0]) * (y - v[k][1]) - (v[(k + 1) % 4][1] - v[k][1]) * (x - v[k][0]))
                   >= 0 for k in range(4)):
                s += float(pred[y, x])
                c += 1
    return s / c if c else 0.0


def unclip(box, ratio=1.5):
    (x0, y0), (x1, y1), _, (x3, y3) = box
    a, b = math.hypot(x1 - x0, y1 - y0), math.hypot(x3 - x0, y3 - y0)
    if a == 0 or b == 0:
        return [list(p) for p in box]
    d = a * b * ratio / (2 * (a + b))
    q = [(float(int(x)), float(int(y))) for x, y in box]
    (x0, y0), (x1, y1), _, (x3, y3) = q
    a, b = math.hypot(x1 - x0, y1 - y0), math.hypot(x3 - x0, y3 - y0)
    if a == 0 or b == 0:
        return [list(p) for p in q]
    u, w = ((x1 - x0) / a, (y1 - y0) / a), ((x3 - x0) / b, (y3 - y0) / b)
    return [[px + d * (su * u[0] + sw * w[0]), py + d * (su * u[1] + sw * w[1])]
            for (px, py), (su, sw) in zip(q, ((-1, -1), (1, -1), (1, 1), (-1, 1)))]


def boxes_from_bitmap(pred, bitmap, dest_width, dest_height, box_thresh=0.7, max_candidates=100, min_size=3,
                      resize=False):
    """pred [H,W] float, bitmap [H,W] bool -> list of boxes [[x,y]*4] (floats holding integers, like the reference)."""
    H, W = bitmap.shape
    boxes = []
    for comp in components(bitmap)[:max_candidates]:
        box, sside = mini_box(comp)
        if sside < min_size:
            continue
        if box_thresh > box_score(pred, box):
            continue
        box, sside = mini_box(unclip(box))
        if sside < min_size + 2:
            continue
        dw, dh = (dest_width, dest_height) if resize else (W, H)
        b = np.array(box, dtype=np.float64)
        b[:, 0] = np.clip(np.round(b[:, 0] / W * dw), 0, dw)
        b[:, 1] = np.clip(np.round(b[:, 1] / H * dh), 0, dh)
        boxes.append(b.tolist())
    return boxes


def synthetic_maps(seed, N=2, H=96, W=128, regions=6):
    """Probability maps with rotated text-like bars (some touching the border, some too small, some weak), plus speckle."""
    g = np.random.default_rng(seed)
    maps = np.zeros((N, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for n in range(N):
        for _ in range(regions):
            cx, cy = g.uniform(0, W), g.uniform(0, H)
            a, b = g.uniform(8, 30), g.uniform(1.5, 7)
            th = g.uniform(-0.8, 0.8)
            u = (xx - cx) * math.cos(th) + (yy - cy) * math.sin(th)
            v = -(xx - cx) * math.sin(th) + (yy - cy) * math.cos(th)
            inside = (np.abs(u) <= a) & (np.abs(v) <= b)
            level = g.choice([0.95, 0.8, 0.55])
            maps[n][inside] = np.maximum(maps[n][inside], level)
        speck = g.uniform(0, 1, (H, W)) < 0.002
        maps[n][speck] = 0.9
        maps[n] += g.uniform(0, 0.05, (H, W)).astype(np.float32)
    return maps
