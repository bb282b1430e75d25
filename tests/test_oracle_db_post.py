"""Anchors of oracle/db_post.py (parity unpinned: cv2 / pyclipper are not installed, see its header) and of the host
geometry the product uses (megreader_amd/structure/db_geometry.py) against it."""
import math

import numpy as np

from megreader_amd.structure import db_geometry as G
from oracle import db_post as O


def test_min_area_rect_of_a_rotated_bar_and_corner_order():
    th, a, b = 0.4, 20.0, 4.0
    c, s = math.cos(th), math.sin(th)
    pts = [(30 + u * c - v * s, 25 + u * s + v * c) for u in (-a, a) for v in (-b, b)] + [(30.0, 25.0)]
    for impl in (O.mini_box, G.mini_box):
        box, sside = impl(pts)
        assert abs(sside - 2 * b) < 1e-9
        sides = sorted(math.dist(box[k], box[(k + 1) % 4]) for k in range(4))
        assert abs(sides[0] - 2 * b) < 1e-9 and abs(sides[3] - 2 * a) < 1e-9
        # reference rule: [0], [3] are the two left-most corners (upper first), [1], [2] the right-most (upper first)
        assert box[0][1] < box[3][1] and box[1][1] < box[2][1]
        assert max(box[0][0], box[3][0]) <= min(box[1][0], box[2][0])


def test_product_geometry_equals_oracle_on_random_point_sets():
    g = np.random.default_rng(0)
    for _ in range(200):
        n = int(g.integers(1, 40))
        pts = [(int(x), int(y)) for x, y in g.integers(0, 50, (n, 2))]
        bo, so = O.mini_box(pts)
        bp, sp = G.mini_box(pts)
        assert abs(so - sp) < 1e-9
        if so > 1e-6 and abs(math.dist(bo[0], bo[1]) - math.dist(bo[1], bo[2])) > 1e-6:   # unique orientation
            assert np.allclose(bo, bp, atol=1e-9)
        assert np.allclose(O.unclip(bo), G.unclip(bo), atol=1e-12)


def test_unclip_grows_every_side_by_area_ratio_over_perimeter():
    box = [[10.0, 10.0], [50.0, 10.0], [50.0, 20.0], [10.0, 20.0]]
    d = 40 * 10 * 1.5 / (2 * 50)
    out = O.unclip(box)
    assert np.allclose(out, [[10 - d, 10 - d], [50 + d, 10 - d], [50 + d, 20 + d], [10 - d, 20 + d]])


def test_components_and_boxes_on_an_axis_aligned_block():
    pred = np.zeros((40, 60), np.float32)
    pred[10:20, 15:45] = 0.9          # 30 x 10 block -> rectangle (29, 9) on pixel centres
    pred[30, 5] = 0.9                 # speckle: short side 0 < min_size
    comps = O.components(pred > 0.3)
    assert [len(c) for c in comps] == [300, 1]
    boxes = O.boxes_from_bitmap(pred, pred > 0.3, 60, 40)
    d = 29 * 9 * 1.5 / (2 * 38)
    want = [[round(15 - d), round(10 - d)], [round(44 + d), round(10 - d)], [round(44 + d), round(19 + d)],
            [round(15 - d), round(19 + d)]]
    assert boxes == [[[float(x), float(y)] for x, y in want]]
    assert O.box_score(pred, [[15, 10], [44, 10], [44, 19], [15, 19]]) == np.float32(0.9)
