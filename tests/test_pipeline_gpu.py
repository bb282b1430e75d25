"""On-device input pipeline (csrc/pipeline.hip, megreader_amd.data.DevicePipeline) against oracle/pipeline.py: resize
(up / down / identity, both modes) + normalise + CHW bit-exact with the numpy restatement of cv2's float32 path, label
encoding identical to charsets.string_to_label, and the prefetcher delivering batches in order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from megreader_amd.charsets import EnglishCharset  # noqa: E402
from megreader_amd.data import DevicePipeline, Prefetcher  # noqa: E402
from oracle.pipeline import process_sample  # noqa: E402


def _samples(seed, n):
    rng = np.random.RandomState(seed)
    shapes = [(32, 128), (31, 100), (48, 200), (20, 37), (64, 256), (17, 300), (33, 33), (100, 40)]
    images = [rng.randint(0, 256, size=shapes[i % len(shapes)] + (3,)).astype(np.uint8) for i in range(n)]
    alphabet = "ABCxyz019 -_Zq"
    texts = ["".join(alphabet[j] for j in rng.randint(0, len(alphabet), size=rng.randint(1, 40))) for _ in range(n)]
    return images, texts


@pytest.mark.parametrize("mode", ["resize", "pad"])
@pytest.mark.parametrize("size", [(32, 128), (64, 256)])
def test_pipeline_matches_oracle(mode, size):
    cs = EnglishCharset()
    pipe = DevicePipeline(image_size=size, mode=mode, charset=cs)
    images, texts = _samples(1, 19)
    batch = pipe.process(images, texts)
    torch.cuda.synchronize()
    assert batch['image'].shape == (19, 3) + size and batch['image'].dtype == torch.float32
    worst = 0.0
    for i, (im, tx) in enumerate(zip(images, texts)):
        chw, lab, ln = process_sample(im, tx, size, mode, cs.index)
        got = batch['image'][i].cpu().numpy()
        worst = max(worst, float(np.abs(got - chw).max()))
        assert np.array_equal(got, chw), (i, im.shape, float(np.abs(got - chw).max()))
        assert np.array_equal(batch['label'][i].cpu().numpy(), lab), (i, tx)
        assert int(batch['length'][i]) == int(ln)


def test_prefetcher_order_and_overlap():
    cs = EnglishCharset()
    pipe = DevicePipeline(image_size=(32, 128), mode='resize', charset=cs)
    batches = [_samples(10 + k, 8) for k in range(5)]
    got = []
    for b in Prefetcher(batches, pipe):
        got.append((b['image'].clone(), b['label'].clone(), b['length'].clone()))
    torch.cuda.synchronize()
    assert len(got) == 5
    for (images, texts), (img, lab, ln) in zip(batches, got):
        for i, (im, tx) in enumerate(zip(images, texts)):
            chw, l_, n_ = process_sample(im, tx, (32, 128), 'resize', cs.index)
            assert np.array_equal(img[i].cpu().numpy(), chw)
            assert np.array_equal(lab[i].cpu().numpy(), l_) and int(ln[i]) == int(n_)
