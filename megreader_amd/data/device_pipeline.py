"""Input pipeline on the GPU.

The reference prepares every sample on the host -- cv2.resize of the float32 image (data/processes/resize_image.py:
29-38), mean subtraction / scaling / HWC->CHW (normalize_image.py:8-17), label encoding (concern/charsets.py:52-58,
make_recognition_label.py:11-24) -- in 2 DataLoader workers (data/data_loader.py:22) and moves the fp32 batch with a
blocking `.to(device)` (structure/model.py:173).  At > 70 k images/s per GPU that is the ceiling of any real run.
Here the host only hands over the DECODED uint8 pixels (4x fewer bytes than the fp32 batch) and the UTF-32 text:

    pipe = DevicePipeline(image_size=(32, 128), mode='resize', charset=charset)
    for batch in Prefetcher(loader_of_(images, texts), pipe):       # H2D copy + kernels of batch i+1 overlap step i
        step.copy_inputs(batch['image'], batch['label'], batch['length']); loss = step()

`process()` = one pinned staging copy (async, side stream) + mr_resize_normalize + mr_encode_labels.
Resize arithmetic follows cv2's float32 INTER_LINEAR path (csrc/pipeline.hip); cv2 itself is not available in the build
image, so that one piece is checked against the numpy restatement in oracle/pipeline.py ("parity unpinned").
"""
import ctypes

import numpy as np
import torch

from .._lib import call, load, ptr
from ..charsets import EnglishCharset

RGB_MEAN = (122.67891434, 116.66876762, 104.00698793)   # data/processes/normalize_image.py:9 (applied to BGR as is)


class ImgDesc(ctypes.Structure):
    """struct ImgDesc of csrc/pipeline.hip."""
    _fields_ = [("offset", ctypes.c_longlong), ("h", ctypes.c_int), ("w", ctypes.c_int), ("pitch", ctypes.c_int),
                ("dst_w", ctypes.c_int), ("scale_x", ctypes.c_double), ("scale_y", ctypes.c_double)]


def charset_table(charset):
    """Sorted (codepoint, id) arrays of a charset, case folding baked in (concern/charsets.py:37-41: `index()`
    upper-cases the query unless case_sensitive).  Multi-codepoint / None entries (blank, unknown) are skipped."""
    pairs = {}
    for i in range(len(charset)):
        ch = charset[i]
        if not isinstance(ch, str) or len(ch) != 1:
            continue
        pairs.setdefault(ord(ch), i)
    if not getattr(charset, "case_sensitive", False):
        for cp, i in list(pairs.items()):
            lo = chr(cp).lower()
            if len(lo) == 1 and lo.upper() == chr(cp):
                pairs.setdefault(ord(lo), i)
    cps = sorted(pairs)
    return np.array(cps, dtype=np.int32), np.array([pairs[c] for c in cps], dtype=np.int32)


def target_width(mode, image_size, shape):
    """_ResizeImage.get_image_size (resize_image.py:40-48)."""
    height, width = image_size
    if mode == 'keep_ratio':
        width = max(width, int(height / shape[0] * shape[1] / 32 + 0.5) * 32)
    if mode == 'pad':
        width = min(width, max(int(height / shape[0] * shape[1] / 32 + 0.5) * 32, 32))
    return width


class DevicePipeline(object):
    def __init__(self, image_size=(32, 128), mode='resize', charset=None, max_size=32, device=None):
        if mode not in ('resize', 'pad'):
            raise NotImplementedError("DevicePipeline supports the batched modes 'resize' and 'pad' "
                                      "(keep_size / keep_ratio produce per-sample shapes)")
        self.image_size = tuple(image_size)
        self.mode = mode
        self.charset = charset if charset is not None else EnglishCharset()
        self.max_size = max_size
        self.device = torch.device(device if device is not None else "cuda")
        load()
        cps, ids = charset_table(self.charset)
        self.tab_cp = torch.from_numpy(cps).to(self.device)
        self.tab_id = torch.from_numpy(ids).to(self.device)
        self.unknown = getattr(self.charset, "unknown", 1)
        self._staging = {}

    def _pinned(self, key, nbytes):
        buf = self._staging.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self._staging[key] = torch.empty((max(nbytes, 1),), dtype=torch.uint8).pin_memory()
        return buf

    def pack(self, images, texts, slot=0):
        """Host side: lay the decoded uint8 HWC images, their descriptors and the UTF-32 text out in ONE pinned
        staging buffer (per prefetch slot).  Returns (pinned uint8 tensor, layout tuple)."""
        n = len(images)
        H, W = self.image_size
        descs = (ImgDesc * n)()
        off = 0
        for i, im in enumerate(images):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise TypeError("images must be uint8 HWC with 3 channels (cv2.imread(..., IMREAD_COLOR))")
            descs[i].offset, descs[i].h, descs[i].w, descs[i].pitch = off, im.shape[0], im.shape[1], im.shape[1] * 3
            descs[i].dst_w = W if self.mode == 'resize' else target_width('pad', self.image_size, im.shape)
            # cv2: inv_scale = (double)dsize / ssize; scale = 1. / inv_scale
            descs[i].scale_x = 1.0 / (float(descs[i].dst_w) / float(im.shape[1]))
            descs[i].scale_y = 1.0 / (float(H) / float(im.shape[0]))
            off += (im.shape[0] * im.shape[1] * 3 + 15) // 16 * 16
        pix_bytes = off
        desc_off = pix_bytes
        desc_bytes = (ctypes.sizeof(ImgDesc) * n + 15) // 16 * 16
        cp = [np.frombuffer(t.encode('utf-32-le'), dtype=np.int32) for t in texts]
        offs = np.zeros(n + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(c) for c in cp])
        text_off = desc_off + desc_bytes
        text_bytes = (int(offs[-1]) * 4 + 15) // 16 * 16
        offs_off = text_off + text_bytes
        total = offs_off + 8 * (n + 1)
        buf = self._pinned(slot, total)
        host = buf.numpy()
        for i, im in enumerate(images):
            nb = im.shape[0] * im.shape[1] * 3
            host[descs[i].offset:descs[i].offset + nb] = np.ascontiguousarray(im).reshape(-1)
        host[desc_off:desc_off + ctypes.sizeof(ImgDesc) * n] = np.frombuffer(bytes(descs), dtype=np.uint8)
        if int(offs[-1]):
            host[text_off:text_off + int(offs[-1]) * 4] = np.concatenate(cp).view(np.uint8)
        host[offs_off:offs_off + 8 * (n + 1)] = offs.view(np.uint8)
        return buf[:total], (n, desc_off, text_off, offs_off)

    def upload(self, staged, layout):
        """One async H2D copy of the staging buffer, then the two kernels, on the CURRENT stream."""
        n, desc_off, text_off, offs_off = layout
        H, W = self.image_size
        dbuf = torch.empty((staged.numel(),), dtype=torch.uint8, device=self.device)
        dbuf.copy_(staged, non_blocking=True)
        image = torch.empty((n, 3, H, W), dtype=torch.float32, device=self.device)
        call("mr_resize_normalize", ptr(dbuf), dbuf.data_ptr() + desc_off, n, H, W, RGB_MEAN[0], RGB_MEAN[1],
             RGB_MEAN[2], ptr(image))
        label = torch.empty((n, self.max_size), dtype=torch.int32, device=self.device)
        length = torch.empty((n,), dtype=torch.int32, device=self.device)
        call("mr_encode_labels", dbuf.data_ptr() + text_off, dbuf.data_ptr() + offs_off, n, self.max_size,
             ptr(self.tab_cp), ptr(self.tab_id), self.tab_cp.numel(), int(self.unknown), ptr(label), ptr(length))
        return {'image': image, 'label': label, 'length': length, '_keepalive': dbuf}

    def process(self, images, texts):
        staged, layout = self.pack(images, texts)
        return self.upload(staged, layout)


class Prefetcher(object):
    """Iterates `(images, texts)` batches from a host loader and yields device batches, keeping one batch in flight:
    the staging copy and the pipeline kernels of batch i+1 run on a side stream while the training step of batch i
    runs on the main stream (replaces the blocking `.to(device)` of structure/model.py:173)."""

    def __init__(self, loader, pipeline):
        self.loader = loader
        self.pipe = pipeline
        self.stream = torch.cuda.Stream(device=pipeline.device)

    def __iter__(self):
        it = iter(self.loader)
        slot = 0
        pending = None

        def launch(item, slot):
            images, texts = item
            staged, layout = self.pipe.pack(images, texts, slot)
            with torch.cuda.stream(self.stream):
                batch = self.pipe.upload(staged, layout)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            return batch, ev

        for item in it:
            nxt = launch(item, slot)
            slot ^= 1
            if pending is not None:
                batch, ev = pending
                torch.cuda.current_stream().wait_event(ev)
                for t in batch.values():
                    t.record_stream(torch.cuda.current_stream())
                yield batch
                # the pinned slot of `pending` is reused two batches later: its H2D copy completed with `ev`
            pending = nxt
        if pending is not None:
            batch, ev = pending
            torch.cuda.current_stream().wait_event(ev)
            for t in batch.values():
                t.record_stream(torch.cuda.current_stream())
            yield batch
