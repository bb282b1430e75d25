"""msgpack sample records -- host-side mirror of the reference's data/unpack_msgpack_data.py:15-52
(`UnpackMsgpackData.convert` / `convert_obj`) that feeds the on-device pipeline.

A record is a msgpack map with byte-string keys; the value under b'img' is an encoded image file (PNG / JPEG), every
other byte string is UTF-8 text, containers recurse.  The reference decodes the image with PIL, converts to RGB and,
in mode 'BGR', swaps the channels with cv2.cvtColor -> float32 HWC for the host-side resize.  Here the decoded image
stays **uint8 HWC** (4x fewer bytes over PCIe) and goes straight into `DevicePipeline.pack`, whose kernels do the
float conversion, resize and normalisation on the GPU; the channel swap is a numpy view (no cv2).

The LMDB container around the records (`image` / `extra` named databases, data/lmdb_dataset.py:59-88) is read by
`megreader_amd.data.lmdb_reader` (pure-Python reader of the on-disk format; parity unpinned, see its header); nori is a
Megvii-internal store and is not read."""
import io

import msgpack
import numpy as np
from PIL import Image


class UnpackMsgpackData(object):
    """`convert(data: bytes) -> dict` with the reference's key / value rules; `mode` 'BGR' (default) or 'RGB'."""

    def __init__(self, mode='BGR'):
        if mode not in ('BGR', 'RGB'):
            raise ValueError("mode must be 'BGR' or 'RGB'")
        self.mode = mode

    def convert_obj(self, obj):
        if isinstance(obj, dict):
            out = {}
            for key, value in obj.items():
                nkey = key.decode() if isinstance(key, bytes) else key
                if nkey == 'img':
                    img = np.array(Image.open(io.BytesIO(value)).convert('RGB'))      # uint8 [H, W, 3]
                    out[nkey] = np.ascontiguousarray(img[:, :, ::-1]) if self.mode == 'BGR' else img
                else:
                    out[nkey] = self.convert_obj(value)
            return out
        if isinstance(obj, list):
            return [self.convert_obj(item) for item in obj]
        if isinstance(obj, bytes):
            return obj.decode()
        return obj

    def convert(self, data):
        # raw=True keeps keys / strings as bytes exactly like the msgpack version the reference was written against
        return self.convert_obj(msgpack.loads(data, raw=True, max_str_len=2 ** 31 - 1, strict_map_key=False))

    def __call__(self, data, data_id=None, meta=None):
        """`data`: the record's bytes (what the reference fetches from LMDB / nori under `data_id`)."""
        meta = {} if meta is None else meta
        item = self.convert(data)
        if data_id is not None:
            item['data_id'] = data_id
        meta.update(item)
        return meta


def records_to_batch(records, pipeline, text_key='gt', mode='BGR'):
    """Decode a list of msgpack records and hand them to a `DevicePipeline`: returns what `pipeline.process` returns
    ({'image': f32 [N,3,H,W] on the GPU, 'label', 'length'})."""
    unpack = UnpackMsgpackData(mode=mode)
    items = [unpack.convert(r) for r in records]
    return pipeline.process([it['img'] for it in items], [it.get(text_key, '') for it in items])
