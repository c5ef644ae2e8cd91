"""Float reference of the LetterBox pixels  --  TEST INFRASTRUCTURE ONLY.

Reference: ``yololite/data/augment.py:636-681`` (LetterBox.__call__): ``cv2.resize(img, new_unpad,
interpolation=cv2.INTER_LINEAR)`` (:666) then ``cv2.copyMakeBorder(..., value=(114, 114, 114))`` (:669-671), followed by
``engine/predictor.py:77-81`` (stack, BGR -> RGB, HWC -> CHW).

cv2 is a third-party dependency that is absent from /root/reference and from this image, and the reference pins no
version: PARITY UNPINNED at the cv2 boundary.  What is restated here is the *published definition* of INTER_LINEAR --
bilinear interpolation with half-pixel centres, ``src = (dst + 0.5) * (src_size / dst_size) - 0.5``, coordinates
clamped to the image, no antialiasing -- evaluated in float64 and rounded to nearest, i.e. WITHOUT OpenCV's 11-bit
fixed-point coefficients.  It is deliberately independent of the product's integer restatement
(``yolo-lite_amd/data/augment.py::resize_linear_u8`` and the HIP kernel ``yl_letterbox_u8``), which must agree with
it to within 1 LSB (the fixed-point rounding); tests/test_letterbox_oracle.py and tests/test_gpu_frontend.py check
both against it.
"""
from __future__ import annotations

import numpy as np

from .yolo11_oracle import letterbox_geometry


def resize_bilinear_f64(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """HWC uint8 -> HWC float64 (not rounded): INTER_LINEAR by its definition."""
    h, w = img.shape[:2]

    def axis(dst, src):
        f = (np.arange(dst, dtype=np.float64) + 0.5) * (src / dst) - 0.5
        f = np.clip(f, 0.0, src - 1.0)
        i0 = np.minimum(np.floor(f).astype(np.int64), src - 1)
        i1 = np.minimum(i0 + 1, src - 1)
        return i0, i1, f - i0

    x0, x1, fx = axis(new_w, w)
    y0, y1, fy = axis(new_h, h)
    s = img.astype(np.float64)
    top = s[y0][:, x0] * (1.0 - fx)[None, :, None] + s[y0][:, x1] * fx[None, :, None]
    bot = s[y1][:, x0] * (1.0 - fx)[None, :, None] + s[y1][:, x1] * fx[None, :, None]
    return top * (1.0 - fy)[:, None, None] + bot * fy[:, None, None]


def letterbox_f64(img: np.ndarray, new_shape=(640, 640), auto: bool = False, stride: int = 32) -> np.ndarray:
    """HWC BGR uint8 -> HWC float64 letterboxed image (augment.py:636-681), border 114."""
    geo = letterbox_geometry(img.shape[:2], new_shape, auto=auto, stride=stride)
    nw, nh = geo["new_unpad"]
    top, bottom, left, right = geo["pads"]
    body = img.astype(np.float64) if (img.shape[1], img.shape[0]) == (nw, nh) else resize_bilinear_f64(img, nw, nh)
    out = np.full((nh + top + bottom, nw + left + right, img.shape[2]), 114.0)
    out[top: top + nh, left: left + nw] = body
    return out


def letterbox_batch_u8(images, new_shape=(640, 640), auto: bool = False, stride: int = 32) -> np.ndarray:
    """list of HWC BGR uint8 -> (B, 3, H, W) uint8 RGB (predictor.py:77-81), float pixels rounded half to even."""
    x = np.stack([np.clip(np.rint(letterbox_f64(im, new_shape, auto, stride)), 0, 255).astype(np.uint8) for im in images])
    return np.ascontiguousarray(x[..., ::-1].transpose(0, 3, 1, 2))
