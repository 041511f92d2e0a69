"""CPU ORACLE (test infrastructure, never a product path) -- image resampling used on the hot path.

PARITY UNPINNED: OpenCV (cv2) is not installed in the build image and is not part of
/root/reference; it is a third-party dependency of the reference (requirements.txt: opencv-python,
unpinned).  These functions restate the published algorithm of cv::resize(INTER_LINEAR) from
OpenCV 4.x `modules/imgproc/src/resize.cpp` (generic, non-IPP path):

  * source coordinate  fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx, with the
    border rule (sx < 0 -> sx = 0, fx = 0; sx >= W-1 -> sx = W-1, fx = 0) horizontally and row
    clamping vertically;
  * uint8: 11-bit fixed point coefficients (INTER_RESIZE_COEF_BITS), horizontal pass to int32,
    vertical pass `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`;
  * float32: horizontal `S[sx]*a0 + S[sx+1]*a1`, vertical `S0*b0 + S1*b1`, all in float32.

Call sites restated: ultralytics/yolov5 v6.0 `utils/augmentations.py::letterbox` (used by
AutoShape, reached from networks/yolo.py:70) and networks/deepsort/deep/feature_extractor.py:36
(`cv2.resize(im.astype(np.float32)/255., (50, 50))`).
"""
from __future__ import annotations

import math

import numpy as np

COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS


def _axis_tables(src, dst):
    """Per-destination-index (source index, fraction) as cv::resize computes them."""
    scale = 1.0 / (float(dst) / float(src))          # double, like `scale_x = 1./inv_scale_x`
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _h_tables(src, dst):
    s, f = _axis_tables(src, dst)
    lo = s < 0
    s = np.where(lo, 0, s)
    f = np.where(lo, np.float32(0), f)
    hi = s >= src - 1
    s = np.where(hi, src - 1, s)
    f = np.where(hi, np.float32(0), f).astype(np.float32)
    s1 = np.minimum(s + 1, src - 1)                  # weight is 0 whenever this clamps
    return s, s1, f


def _v_tables(src, dst):
    s, f = _axis_tables(src, dst)
    return np.clip(s, 0, src - 1), np.clip(s + 1, 0, src - 1), f.astype(np.float32)


def _round_half_even_i(x):
    return np.rint(x).astype(np.int64)               # cvRound == lrint (nearest-even)


def resize_linear_u8(img, dst_w, dst_h):
    """cv2.resize(img_u8, (dst_w, dst_h), interpolation=INTER_LINEAR)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    if (w, h) == (dst_w, dst_h):
        return img.copy()
    x0, x1, fx = _h_tables(w, dst_w)
    y0, y1, fy = _v_tables(h, dst_h)
    a0 = np.clip(_round_half_even_i((np.float32(1) - fx) * np.float32(COEF_ONE)), -32768, 32767)
    a1 = np.clip(_round_half_even_i(fx * np.float32(COEF_ONE)), -32768, 32767)
    b0 = np.clip(_round_half_even_i((np.float32(1) - fy) * np.float32(COEF_ONE)), -32768, 32767)
    b1 = np.clip(_round_half_even_i(fy * np.float32(COEF_ONE)), -32768, 32767)
    src = img.astype(np.int64)
    hrow = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]   # (h, dst_w, c) int
    s0 = hrow[y0]
    s1 = hrow[y1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_linear_f32(img, dst_w, dst_h):
    """cv2.resize(img_f32, (dst_w, dst_h)) (default INTER_LINEAR) on float32 HxWxC."""
    img = np.asarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    if (w, h) == (dst_w, dst_h):
        return img.copy()
    x0, x1, fx = _h_tables(w, dst_w)
    y0, y1, fy = _v_tables(h, dst_h)
    a0 = (np.float32(1) - fx).astype(np.float32)[None, :, None]
    a1 = fx[None, :, None]
    hrow = (img[:, x0, :] * a0).astype(np.float32) + (img[:, x1, :] * a1).astype(np.float32)
    hrow = hrow.astype(np.float32)
    b0 = (np.float32(1) - fy).astype(np.float32)[:, None, None]
    b1 = fy[:, None, None]
    out = (hrow[y0] * b0).astype(np.float32) + (hrow[y1] * b1).astype(np.float32)
    return out.astype(np.float32)


def py_round(x):
    """Python 3 round() (banker's), as used by letterbox."""
    return int(round(x))


def letterbox_geometry(h0, w0, new_h, new_w):
    """yolov5 v6.0 letterbox(auto=False, scaleFill=False, scaleup=True): resize size and pads."""
    r = min(new_h / h0, new_w / w0)
    unpad_w, unpad_h = py_round(w0 * r), py_round(h0 * r)
    dw, dh = (new_w - unpad_w) / 2, (new_h - unpad_h) / 2
    top, bottom = py_round(dh - 0.1), py_round(dh + 0.1)
    left, right = py_round(dw - 0.1), py_round(dw + 0.1)
    return unpad_w, unpad_h, top, bottom, left, right


def letterbox(img, new_h, new_w, color=114):
    h0, w0 = img.shape[:2]
    uw, uh, top, bottom, left, right = letterbox_geometry(h0, w0, new_h, new_w)
    im = resize_linear_u8(img, uw, uh) if (w0, h0) != (uw, uh) else img
    out = np.full((uh + top + bottom, uw + left + right, 3), color, dtype=np.uint8)
    out[top:top + uh, left:left + uw] = im
    return out


def autoshape_size(shapes_hw, size=640, stride=32):
    """AutoShape.forward: common inference shape for a list of (h, w) (models/common.py v6.0)."""
    s1 = []
    for (h, w) in shapes_hw:
        g = size / max(h, w)
        s1.append([h * g, w * g])
    mx = np.stack(s1, 0).max(0)
    return [int(math.ceil(x / stride) * stride) for x in mx]
