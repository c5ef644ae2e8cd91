"""LetterBox for inference (reference ``yololite/data/augment.py:612-681``).

The geometry (ratio, unpadded size, stride-modulo padding when ``auto``, the ``round(d -/+ 0.1)``
split, border value 114) follows the reference exactly and is pinned by golden vectors.  The pixels
are not pinnable: the reference delegates them to ``cv2.resize(INTER_LINEAR)`` /
``cv2.copyMakeBorder``, a third-party library that is absent here.  ``resize_linear_u8`` restates
OpenCV's published 8-bit bilinear algorithm (half-pixel centres, no antialiasing, 11-bit fixed-point
coefficients) so results should agree with cv2 to within rounding, but that is unverified.
"""
from __future__ import annotations

import numpy as np

_COEF_BITS = 11
_COEF_ONE = 1 << _COEF_BITS


def _axis_coeffs(dst: int, src: int):
    scale = src / dst
    fx = (np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5
    fx = fx.astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx
    lo = sx < 0
    fx[lo], sx[lo] = 0.0, 0
    hi = sx >= src - 1
    fx[hi], sx[hi] = 0.0, src - 1
    c1 = np.rint(fx * _COEF_ONE).astype(np.int32)
    c0 = np.rint((1.0 - fx) * _COEF_ONE).astype(np.int32)
    return sx, np.minimum(sx + 1, src - 1), c0, c1


def resize_linear_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """HWC uint8 bilinear resize in OpenCV's INTER_LINEAR fixed-point convention."""
    h, w = img.shape[:2]
    x0, x1, a0, a1 = _axis_coeffs(new_w, w)
    y0, y1, b0, b1 = _axis_coeffs(new_h, h)
    src = img.astype(np.int32)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]  # (h, new_w, c), scaled by 2^11
    s0, s1 = rows[y0] >> 4, rows[y1] >> 4
    out = (((b0[:, None, None] * s0) >> 16) + ((b1[:, None, None] * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class LetterBox:
    """Resize keeping aspect ratio and pad to ``new_shape`` (or to a stride multiple when ``auto``)."""

    def __init__(self, new_shape=(640, 640), auto=False, scaleFill=False, scaleup=True, center=True, stride=32):
        self.new_shape = new_shape
        self.auto = auto
        self.scaleFill = scaleFill
        self.scaleup = scaleup
        self.stride = stride
        self.center = center

    def geometry(self, shape):
        """(new_unpad (w, h), (top, bottom, left, right), ratio) for an input (h, w)."""
        new_shape = self.new_shape
        if isinstance(new_shape, int):
            new_shape = (new_shape, new_shape)
        r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
        if not self.scaleup:
            r = min(r, 1.0)
        ratio = r, r
        new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
        dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
        if self.auto:
            dw, dh = dw % int(self.stride), dh % int(self.stride)
        elif self.scaleFill:
            dw, dh = 0.0, 0.0
            new_unpad = (new_shape[1], new_shape[0])
            ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
        if self.center:
            dw /= 2
            dh /= 2
        top, bottom = (int(round(dh - 0.1)) if self.center else 0), int(round(dh + 0.1))
        left, right = (int(round(dw - 0.1)) if self.center else 0), int(round(dw + 0.1))
        return new_unpad, (top, bottom, left, right), ratio

    def __call__(self, labels=None, image=None):
        img = image if image is not None else (labels or {}).get("img")
        shape = img.shape[:2]
        new_unpad, (top, bottom, left, right), _ = self.geometry(shape)
        if shape[::-1] != new_unpad:
            img = resize_linear_u8(img, new_unpad[0], new_unpad[1])
        out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), 114, dtype=img.dtype)
        out[top: top + img.shape[0], left: left + img.shape[1]] = img
        return out


class DeviceLetterBox:
    """LetterBox + stack + BGR->RGB + HWC->CHW on the GPU (``yl_letterbox_u8``): the raw uint8 source images
    cross PCIe (fewer bytes than the letterboxed float batch) and one kernel per image writes its slot of the
    uint8 NCHW batch the network's first layer reads.  Geometry = ``LetterBox.geometry``; pixels = the same
    fixed-point arithmetic as ``resize_linear_u8`` (bit-identical, tests/test_gpu_frontend.py)."""

    def __init__(self, new_shape=(640, 640), auto=False, stride=32, device="cuda:0"):
        import torch

        self.lb = LetterBox(new_shape, auto=auto, stride=stride)
        self.device = torch.device(device)
        self._tabs = {}

    def _tab(self, dst: int, src: int):
        import torch

        key = (dst, src)
        if key not in self._tabs:
            i0, i1, c0, c1 = _axis_coeffs(dst, src)
            t = np.stack([i0, i1, c0, c1], axis=1).astype(np.int32)
            self._tabs[key] = torch.from_numpy(np.ascontiguousarray(t)).to(self.device)
        return self._tabs[key]

    def __call__(self, images):
        """list of HWC BGR uint8 arrays (all mapping to one output shape) -> (B, 3, H, W) uint8 tensor on the device."""
        import torch

        from .. import _lib

        L = _lib.lib()
        geos = [self.lb.geometry(im.shape[:2]) for im in images]
        shapes = {(nu[1] + p[0] + p[1], nu[0] + p[2] + p[3]) for nu, p, _ in geos}
        if len(shapes) != 1:
            raise ValueError(f"images letterbox to different shapes {sorted(shapes)}; batch them by shape")
        H, W = shapes.pop()
        out = torch.empty((len(images), 3, H, W), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        keep = []
        for b, (im, (new_unpad, (top, _bottom, left, _right), _)) in enumerate(zip(images, geos)):
            if im.ndim != 3 or im.shape[2] != 3 or im.dtype != np.uint8:
                raise TypeError("DeviceLetterBox takes HWC uint8 images with 3 channels")
            src = torch.from_numpy(np.ascontiguousarray(im)).to(self.device, non_blocking=True)
            keep.append(src)
            h, w = im.shape[:2]
            nw, nh = new_unpad
            resize = (w, h) != (nw, nh)
            xt = self._tab(nw, w) if resize else None
            yt = self._tab(nh, h) if resize else None
            _lib.check(L.yl_letterbox_u8(src.data_ptr(), h, w, 3 * w, out[b].data_ptr(), H, W, top, left, nh, nw,
                                         xt.data_ptr() if resize else None, yt.data_ptr() if resize else None, stream),
                       "yl_letterbox_u8")
        for t in keep:  # the uploads must outlive the kernels queued on this stream
            t.record_stream(torch.cuda.current_stream(self.device))
        return out
