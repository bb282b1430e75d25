#!/usr/bin/env python
"""Find where the GPU resize+normalise differs from oracle/pipeline.py and which arithmetic variant it matches."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megreader_amd.charsets import EnglishCharset  # noqa: E402
from megreader_amd.data import DevicePipeline  # noqa: E402
from oracle.pipeline import RGB_MEAN, _taps, process_sample, resize_linear_f32  # noqa: E402

rng = np.random.RandomState(1)
shapes = [(32, 128), (31, 100), (48, 200), (20, 37), (64, 256), (17, 300), (33, 33), (100, 40)]
images = [rng.randint(0, 256, size=shapes[i % len(shapes)] + (3,)).astype(np.uint8) for i in range(8)]
cs = EnglishCharset()
for mode in ("pad", "resize"):
    pipe = DevicePipeline(image_size=(32, 128), mode=mode, charset=cs)
    batch = pipe.process(images, ["A"] * len(images))
    torch.cuda.synchronize()
    for i, im in enumerate(images):
        chw, _, _ = process_sample(im, "A", (32, 128), mode, cs.index)
        got = batch['image'][i].cpu().numpy()
        bad = np.argwhere(got != chw)
        print(mode, i, im.shape, "mismatches:", len(bad))
        for c, y, x in bad[:3]:
            w_t = 128 if mode == "resize" else min(128, max(int(32 / im.shape[0] * im.shape[1] / 32 + 0.5) * 32, 32))
            sx, fx = _taps(w_t, im.shape[1])
            sy, fy = _taps(32, im.shape[0])
            f = im.astype(np.float32)
            p00, p01 = f[sy[y], sx[x], c], f[sy[y], min(sx[x] + 1, im.shape[1] - 1), c]
            y1 = min(sy[y] + 1, im.shape[0] - 1)
            p10, p11 = f[y1, sx[x], c], f[y1, min(sx[x] + 1, im.shape[1] - 1), c]
            a1, b1 = fx[x], fy[y]
            a0, b0 = np.float32(1) - a1, np.float32(1) - b1
            top = np.float32(np.float32(p00 * a0) + np.float32(p01 * a1))
            bot = np.float32(np.float32(p10 * a0) + np.float32(p11 * a1))
            v = np.float32(np.float32(top * b0) + np.float32(bot * b1))
            n1 = np.float32(np.float32(np.float64(v) - RGB_MEAN[c]) / np.float32(255))
            n2 = np.float32(np.float32(np.float64(v) - RGB_MEAN[c]) * np.float32(1.0 / 255.0))
            print("   at", (c, y, x), "gpu %.9g oracle %.9g | recomputed div %.9g mul-recip %.9g | v=%.9g fx=%.9g fy=%.9g"
                  % (got[c, y, x], chw[c, y, x], n1, n2, v, a1, b1))
