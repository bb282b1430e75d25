"""Oracle restatement of the reference's host input pipeline for one sample (numpy):

  data/processes/resize_image.py:29-53   cv2.resize(image_float32, (width, height))  -- INTER_LINEAR, modes resize / pad
  data/processes/normalize_image.py:8-17 image -= RGB_MEAN (float64 array -> computed in double, stored f32);
                                         image /= 255. (f32); HWC -> CHW
  concern/charsets.py:37-58, data/processes/make_recognition_label.py:11-24   label / length

PARITY UNPINNED for the resize arithmetic: cv2 (opencv-python, unpinned in requirement.txt) is not installed in the build
image, so `resize_linear_f32` restates OpenCV's published float32 INTER_LINEAR algorithm (imgproc/src/resize.cpp:
scale = 1 / (dsize / ssize) in double; fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); taps clamped to the
image with the out-of-range tap's weight forced to 0; horizontal pass then vertical pass in float) and the GPU kernel is
checked against this restatement.  Normalisation and label encoding are plain numpy / python and are exact.
"""
import numpy as np

RGB_MEAN = np.array([122.67891434, 116.66876762, 104.00698793])


def _taps(dst, src):
    scale = 1.0 / (float(dst) / float(src))
    idx = np.zeros(dst, dtype=np.int64)
    frac = np.zeros(dst, dtype=np.float32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0)
        idx[d], frac[d] = s, f
    return idx, frac


def resize_linear_f32(image, width, height):
    """image: float32 HWC.  Returns float32 [height, width, C]."""
    img = np.asarray(image, dtype=np.float32)
    h, w = img.shape[:2]
    if (h, w) == (height, width):
        return img.copy()
    sx, fx = _taps(width, w)
    sy, fy = _taps(height, h)
    sx1 = np.minimum(sx + 1, w - 1)
    sy1 = np.minimum(sy + 1, h - 1)
    a0 = (np.float32(1) - fx)[None, :, None]
    a1 = fx[None, :, None]
    rows = (img[:, sx] * a0).astype(np.float32) + (img[:, sx1] * a1).astype(np.float32)      # horizontal pass
    rows = rows.astype(np.float32)
    b0 = (np.float32(1) - fy)[:, None, None]
    b1 = fy[:, None, None]
    out = (rows[sy] * b0).astype(np.float32) + (rows[sy1] * b1).astype(np.float32)           # vertical pass
    return out.astype(np.float32)


def process_sample(image_u8, text, image_size=(32, 128), mode='resize', charset_index=None, max_size=32):
    """One sample through ResizeImage -> NormalizeImage -> MakeRecognitionLabel.  Returns (image f32 CHW, label i32
    [max_size], length)."""
    height, width = image_size
    img = image_u8.astype('float32')                        # data/file_dataset.py:56
    if mode == 'pad':
        w_t = min(width, max(int(height / img.shape[0] * img.shape[1] / 32 + 0.5) * 32, 32))
        canvas = np.zeros((height, width, 3), np.float32)
        canvas[:, :w_t, :] = resize_linear_f32(img, w_t, height)
        img = canvas
    else:
        img = resize_linear_f32(img, width, height)
    img = (img.astype(np.float64) - RGB_MEAN).astype(np.float32)      # in-place f32 -= f64 array: computed in double
    img = (img / np.float32(255.)).astype(np.float32)
    chw = np.ascontiguousarray(img.transpose(2, 0, 1))
    length = max(max_size, len(text))
    target = np.zeros((length,), dtype=np.int32)
    for i, c in enumerate(text):
        target[i] = charset_index(c)
    return chw, target[:max_size], np.int32(min(len(text), max_size))
