"""DB detector post-processing -- mirror of structure/representers/seg_detector_representer.py:10-168
(`SegDetectorRepresenter`: `represent`, `binarize`, `boxes_from_bitmap`, `box_score_fast`, `unclip`, `get_mini_boxes`)
with the per-pixel stages on the GPU (csrc/db_post.hip: connected components + hull candidates, box scores) and the
per-box geometry on the host (db_geometry.py).  Same constructor states, `represent(batch, pred)` contract and box
format (a list per image of [[x, y] * 4] lists, corners ordered top-left, top-right, bottom-right, bottom-left).

Differences, all forced by the build image having neither cv2 nor pyclipper to run the reference against ("parity
unpinned"; the restatement in oracle/db_post.py is the checker):
  * cv2.findContours(RETR_LIST) also returns the contours of HOLES; here a candidate is an 8-connected component (its
    outer contour).  Candidates are taken in raster order of their first pixel, the reference in cv2's order -- it only
    matters when an image has more than `max_candidates` components.
  * rectangles come from exact float64 calipers (cv2 works in float32), box scores count pixels inside or on the border
    of the integer-truncated box (cv2.fillPoly's scan conversion), unclip is the closed form for rectangles."""
import numpy as np
import torch

from .._lib import call, ptr, require_cuda
from .db_geometry import mini_box, unclip


class SegDetectorRepresenter(object):
    def __init__(self, thresh=0.3, box_thresh=0.7, max_candidates=100, resize=False, dest='binary', cmd={}, **kwargs):
        self.thresh, self.box_thresh, self.max_candidates = thresh, box_thresh, max_candidates
        self.resize, self.dest = resize, dest
        self.min_size = 3
        self.scale_ratio = 0.4
        self.debug = cmd.get('debug', False)
        for k in ('thresh', 'box_thresh', 'dest'):
            if k in cmd:
                setattr(self, k, cmd[k])

    def represent(self, batch, _pred):
        """batch['image'] (N,C,H,W), batch['shape'][i] = (height, width) of the original image; _pred[dest] (N,1,H,W).
        Returns (boxes_batch, _pred) like the reference."""
        pred = _pred[self.dest]
        shapes = [tuple(int(v) for v in s) for s in batch['shape']]
        boxes_batch = self.boxes_from_maps(_pred['binary'], pred, shapes)
        return boxes_batch, _pred

    def binarize(self, pred):
        return pred > self.thresh

    def boxes_from_bitmap(self, pred, _bitmap, dest_width, dest_height):
        """Single map, reference signature: pred (1,H,W) probabilities, _bitmap (1,H,W) binarised map."""
        assert _bitmap.size(0) == 1
        boxes = self.boxes_from_maps(pred.reshape(1, 1, *pred.shape[-2:]), None, [(dest_height, dest_width)],
                                     bitmap=_bitmap.reshape(1, 1, *_bitmap.shape[-2:]))[0]
        return boxes, _bitmap[0]

    def boxes_from_maps(self, binary, dest_map, shapes, bitmap=None):
        """binary: the probability map scored for box confidence (N,1,H,W); dest_map: the map that is thresholded into
        regions (defaults to `binary`); bitmap: an already binarised map instead of dest_map."""
        require_cuda(binary)
        prob = binary.detach().float().contiguous()
        N, _, H, W = prob.shape
        if bitmap is not None:
            seg, thr = bitmap.detach().float().contiguous(), 0.5
        else:
            seg, thr = (prob if dest_map is None else dest_map.detach().float().contiguous()), float(self.thresh)
        dev = prob.device
        labels = torch.empty((N, H, W), dtype=torch.int32, device=dev)
        cap = max(4096, N * H * W // 8)
        while True:
            points = torch.empty((cap, 4), dtype=torch.int32, device=dev)
            count = torch.zeros((1,), dtype=torch.int32, device=dev)
            call("mr_db_components", ptr(seg), thr, ptr(labels), ptr(points), ptr(count), cap, N, H, W)
            k = int(count.item())
            if k <= cap:
                break
            cap = k
        pts = points[:k].cpu().numpy()
        # group the hull candidates by (image, component root); raster order of the roots
        comps = {}
        for n, root, x, y in pts.tolist():
            comps.setdefault((n, root), []).append((x, y))
        per_image = [[] for _ in range(N)]
        for (n, root) in sorted(comps):
            per_image[n].append(comps[(n, root)])
        cand = []          # (image, ordered box)
        for n in range(N):
            for contour in per_image[n][:self.max_candidates]:
                box, sside = mini_box(contour)
                if sside < self.min_size:
                    continue
                cand.append((n, box))
        boxes_batch = [[] for _ in range(N)]
        if not cand:
            return boxes_batch
        rows = np.array([[n] + [float(int(v)) for p in box for v in p] for n, box in cand], dtype=np.float32)
        bx = torch.from_numpy(rows).to(dev)
        out = torch.empty((len(cand), 2), dtype=torch.float32, device=dev)
        call("mr_db_box_scores", ptr(prob), ptr(bx), ptr(out), len(cand), N, H, W)
        sc = out.cpu().numpy()
        for (n, box), (s, c) in zip(cand, sc.tolist()):
            score = s / c if c > 0 else 0.0
            if self.box_thresh > score:
                continue
            box2, sside = mini_box(unclip(box))
            if sside < self.min_size + 2:
                continue
            dest_height, dest_width = shapes[n] if self.resize else (H, W)
            b = np.array(box2, dtype=np.float64)
            b[:, 0] = np.clip(np.round(b[:, 0] / W * dest_width), 0, dest_width)
            b[:, 1] = np.clip(np.round(b[:, 1] / H * dest_height), 0, dest_height)
            boxes_batch[n].append(b.tolist())
        return boxes_batch
