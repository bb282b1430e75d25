"""Host-side geometry of the DB detector's post-processing (a few hundred points per image): convex hull, minimum-area
rectangle (what cv2.minAreaRect + cv2.boxPoints return for a contour), the reference's corner ordering
(`get_mini_boxes`, structure/representers/seg_detector_representer.py:128-148) and its `unclip` (:120-126).

cv2 / pyclipper / shapely are not dependencies: the rectangle comes from exact rotating calipers in float64, and the
unclip of a RECTANGLE by distance d with round joins followed by another minAreaRect (what the reference does) is the
rectangle grown by d on every side."""
import math


def convex_hull(points):
    """Andrew's monotone chain; `points` iterable of (x, y); returns the hull counter-clockwise (y up), no duplicates."""
    pts = sorted(set((float(x), float(y)) for x, y in points))
    if len(pts) <= 2:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return lower[:-1] + upper[:-1]


def min_area_rect(points):
    """-> (corners [4][2] in order around the rectangle, (side_a, side_b)).  One side of the optimal rectangle is
    collinear with a hull edge (rotating calipers); ties keep the first edge."""
    hull = convex_hull(points)
    if len(hull) == 1:
        x, y = hull[0]
        return [[x, y]] * 4, (0.0, 0.0)
    if len(hull) == 2:
        (x0, y0), (x1, y1) = hull
        return [[x0, y0], [x1, y1], [x1, y1], [x0, y0]], (math.hypot(x1 - x0, y1 - y0), 0.0)
    best = None
    n = len(hull)
    for i in range(n):
        x0, y0 = hull[i]
        x1, y1 = hull[(i + 1) % n]
        ex, ey = x1 - x0, y1 - y0
        ln = math.hypot(ex, ey)
        if ln == 0.0:
            continue
        ux, uy = ex / ln, ey / ln
        lo_u = hi_u = lo_v = hi_v = None
        for px, py in hull:
            pu = (px - x0) * ux + (py - y0) * uy
            pv = -(px - x0) * uy + (py - y0) * ux
            lo_u = pu if lo_u is None or pu < lo_u else lo_u
            hi_u = pu if hi_u is None or pu > hi_u else hi_u
            lo_v = pv if lo_v is None or pv < lo_v else lo_v
            hi_v = pv if hi_v is None or pv > hi_v else hi_v
        area = (hi_u - lo_u) * (hi_v - lo_v)
        if best is None or area < best[0] - 1e-12:
            best = (area, x0, y0, ux, uy, lo_u, hi_u, lo_v, hi_v)
    _, x0, y0, ux, uy, lo_u, hi_u, lo_v, hi_v = best
    corners = [[x0 + a * ux - b * uy, y0 + a * uy + b * ux] for a, b in
               ((lo_u, lo_v), (hi_u, lo_v), (hi_u, hi_v), (lo_u, hi_v))]
    return corners, (hi_u - lo_u, hi_v - lo_v)


def mini_box(points):
    """`get_mini_boxes`: corners ordered top-left, top-right, bottom-right, bottom-left by the reference's rule (sort by x;
    of the two left-most the upper one first, of the two right-most the upper one second) + the short side."""
    corners, sides = min_area_rect(points)
    p = sorted(corners, key=lambda c: c[0])
    i1, i4 = (0, 1) if p[1][1] > p[0][1] else (1, 0)
    i2, i3 = (2, 3) if p[3][1] > p[2][1] else (3, 2)
    return [p[i1], p[i2], p[i3], p[i4]], min(sides)


def unclip(box, ratio=1.5):
    """Reference: distance = area * 1.5 / perimeter of the box, pyclipper round-join offset of the (integer-truncated)
    box by it.  For a rectangle that offset is the rounded rectangle inscribed in the box grown by `distance` on every
    side, which is what the following minAreaRect recovers."""
    (x0, y0), (x1, y1), (x2, y2), (x3, y3) = box
    a, b = math.hypot(x1 - x0, y1 - y0), math.hypot(x3 - x0, y3 - y0)
    if a == 0.0 or b == 0.0:
        return [list(p) for p in box]
    d = a * b * ratio / (2.0 * (a + b))
    q = [(float(int(x)), float(int(y))) for x, y in box]           # pyclipper works on integer coordinates
    (x0, y0), (x1, y1), _, (x3, y3) = q
    a, b = math.hypot(x1 - x0, y1 - y0), math.hypot(x3 - x0, y3 - y0)
    if a == 0.0 or b == 0.0:
        return [list(p) for p in q]
    ux, uy, vx, vy = (x1 - x0) / a, (y1 - y0) / a, (x3 - x0) / b, (y3 - y0) / b
    out = []
    for (px, py), (su, sv) in zip(q, ((-1, -1), (1, -1), (1, 1), (-1, 1))):
        out.append([px + d * (su * ux + sv * vx), py + d * (su * uy + sv * vy)])
    return out
