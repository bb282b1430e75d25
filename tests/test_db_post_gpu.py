"""DB detector post-processing on the GPU: megreader_amd.structure.SegDetectorRepresenter
(csrc/db_post.hip: union-find connected components + hull candidates, box scores; host geometry) against
oracle/db_post.py on synthetic probability maps -- rotated bars, bars cut by the border, specks, weak regions.
Bit-for-bit: component labels / sizes (integer work); boxes: identical lists (coordinates are rounded integers; the
float64 geometry is the same arithmetic on both sides, the GPU sums of the score are f32 over < 2000 pixels, compared
to the oracle's float64 mean only through the threshold -- maps keep scores away from box_thresh)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from megreader_amd._lib import call, ptr  # noqa: E402
from megreader_amd.structure import SegDetectorRepresenter  # noqa: E402
from oracle import db_post as O  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_components_match_flood_fill(seed):
    maps = O.synthetic_maps(seed, N=2, H=64, W=80)
    prob = torch.from_numpy(maps).to(DEV)
    N, H, W = maps.shape
    labels = torch.empty((N, H, W), dtype=torch.int32, device=DEV)
    points = torch.empty((N * H * W, 4), dtype=torch.int32, device=DEV)
    count = torch.zeros((1,), dtype=torch.int32, device=DEV)
    call("mr_db_components", ptr(prob), 0.3, ptr(labels), ptr(points), ptr(count), N * H * W, N, H, W)
    lab = labels.cpu().numpy()
    pts = points[:int(count.item())].cpu().numpy()
    for n in range(N):
        comps = O.components(maps[n] > 0.3)
        assert ((lab[n] >= 0) == (maps[n] > 0.3)).all()
        roots = set()
        for comp in comps:
            ids = {int(lab[n][y, x]) for x, y in comp}
            assert len(ids) == 1, "a component carries more than one label"
            root = ids.pop()
            fx, fy = min(comp, key=lambda p: (p[1], p[0]))
            assert root == fy * W + fx, "the label is the raster-first pixel of the component"
            roots.add(root)
            mine = {(int(x), int(y)) for nn, r, x, y in pts.tolist() if nn == n and r == root}
            ends = {(x, y) for x, y in comp if x == 0 or x == W - 1 or not maps[n][y, x - 1] > 0.3
                    or not maps[n][y, x + 1] > 0.3}
            assert mine == ends, "run end points"
        assert len(roots) == len(comps)


@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_boxes_match_oracle(seed):
    maps = O.synthetic_maps(seed, N=3, H=96, W=128, regions=7)
    rep = SegDetectorRepresenter(thresh=0.3, box_thresh=0.7, max_candidates=100)
    pred = {'binary': torch.from_numpy(maps).to(DEV).unsqueeze(1)}
    batch = {'image': torch.empty(3, 3, 96, 128), 'shape': [(96, 128)] * 3}
    boxes_batch, out = rep.represent(batch, pred)
    assert out is pred and len(boxes_batch) == 3
    total = 0
    for n in range(3):
        want = O.boxes_from_bitmap(maps[n], maps[n] > 0.3, 128, 96)
        assert boxes_batch[n] == want, (n, boxes_batch[n], want)
        total += len(want)
    assert total >= 3, "the synthetic maps must produce boxes"
    # resize=True scales to the original image shape (seg_detector_representer.py:106-114)
    rep2 = SegDetectorRepresenter(resize=True)
    b2, _ = rep2.represent({'image': batch['image'], 'shape': [(192, 384)] * 3}, pred)
    for n in range(3):
        assert b2[n] == O.boxes_from_bitmap(maps[n], maps[n] > 0.3, 384, 192, resize=True)


def test_reference_signature_boxes_from_bitmap_and_empty_map():
    maps = O.synthetic_maps(9, N=1, H=64, W=64)
    rep = SegDetectorRepresenter()
    pred = torch.from_numpy(maps).to(DEV)
    boxes, bitmap = rep.boxes_from_bitmap(pred, rep.binarize(pred), 64, 64)
    assert boxes == O.boxes_from_bitmap(maps[0], maps[0] > 0.3, 64, 64)
    empty = torch.zeros(1, 1, 32, 32, device=DEV)
    assert rep.represent({'image': empty, 'shape': [(32, 32)]}, {'binary': empty})[0] == [[]]
    with pytest.raises(NotImplementedError):
        rep.represent({'image': empty.cpu(), 'shape': [(32, 32)]}, {'binary': empty.cpu()})
