"""Synthetic device-independent batches of the recognition batch contract (SURVEY.md §8 b5 / BASELINE.md §3):

    {'image': f32 [N,3,H,W] = (uint8 pixel - RGB_MEAN) / 255   (data/processes/normalize_image.py:8-16),
     'label': i32 [N,32] class ids in 2..C-1, zero padded      (data/processes/make_recognition_label.py:11-24),
     'length': i32 [N]}

Used by bench.py and the examples; there is no dataset on the benchmark box.  (tests/test_host_logic_cpu.py checks that
the oracle's own generator, which the golden fixtures were made with, yields the same tensors.)
"""
import torch

RGB_MEAN = (122.67891434, 116.66876762, 104.00698793)


def recognition_batch(n, height=32, width=128, seed=0, max_label=32, min_len=3, max_len=10, num_classes=38):
    """CRNN / attention workloads: label length ~ U{min_len..max_len}."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.randint(0, 256, (n, height, width, 3), generator=g, dtype=torch.int64).float()
    image = ((pix - torch.tensor(RGB_MEAN)) / 255.0).permute(0, 3, 1, 2).contiguous()
    length = torch.randint(min_len, max_len + 1, (n,), generator=g, dtype=torch.int64)
    label = torch.zeros((n, max_label), dtype=torch.int64)
    for i in range(n):
        label[i, :length[i]] = torch.randint(2, num_classes, (int(length[i]),), generator=g)
    return {'image': image, 'label': label.int(), 'length': length.int()}


def recognition_batch_2d(n, height=32, width=64, seed=0, max_label=32, max_len=3, num_classes=38):
    """2D-CTC workload: labels short enough for the W/8 time steps of the head (L + repeats <= T)."""
    return recognition_batch(n, height, width, seed, max_label, 1, max_len, num_classes)


def detection_batch(n, size=640, seed=0, boxes=12):
    """DB detector workload (experiments/seg_detector/seg_detector_db.yaml): image f32 [N,3,S,S]
    normalised like the recognition crops, `gt` [N,1,S,S] in {0,1} (random text rectangles), `mask` [N,S,S] (1 = train
    on this pixel), `thresh_map` [N,S,S] in [0.3, 0.7] on a border band around every rectangle, `thresh_mask` [N,S,S]."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.randint(0, 256, (n, size, size, 3), generator=g, dtype=torch.int64).float()
    image = ((pix - torch.tensor(RGB_MEAN)) / 255.0).permute(0, 3, 1, 2).contiguous()
    gt = torch.zeros((n, 1, size, size))
    thresh_map = torch.zeros((n, size, size))
    thresh_mask = torch.zeros((n, size, size))
    for i in range(n):
        for _ in range(boxes):
            w = int(torch.randint(40, 200, (1,), generator=g))
            h = int(torch.randint(12, 48, (1,), generator=g))
            x0 = int(torch.randint(8, size - w - 8, (1,), generator=g))
            y0 = int(torch.randint(8, size - h - 8, (1,), generator=g))
            gt[i, 0, y0:y0 + h, x0:x0 + w] = 1.0
            thresh_mask[i, y0 - 4:y0 + h + 4, x0 - 4:x0 + w + 4] = 1.0
            thresh_map[i, y0 - 4:y0 + h + 4, x0 - 4:x0 + w + 4] = 0.3
            thresh_map[i, y0:y0 + h, x0:x0 + w] = 0.7
    mask = torch.ones((n, size, size))
    return {'image': image, 'gt': gt, 'mask': mask, 'thresh_map': thresh_map, 'thresh_mask': thresh_mask}
