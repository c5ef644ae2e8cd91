"""Post-processing ops with the reference's names and argument meaning (``yololite/utils/ops.py``).

``non_max_suppression`` runs entirely on the GPU through ``yl_nms`` (no CPU fallback).  The small
box helpers (``scale_boxes`` / ``clip_boxes`` / ``xywh2xyxy`` / ``xyxy2xywh``) are tensor one-liners
kept for API compatibility; the fused path applies ``scale_boxes`` inside the NMS kernel.
"""
from __future__ import annotations

import contextlib
import time
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import _lib


class Profile(contextlib.ContextDecorator):
    """Wall-clock stage timer; synchronises the device when it is a GPU (reference ops.py:18-63)."""

    def __init__(self, t: float = 0.0, device: Optional[torch.device] = None):
        self.t = t
        self.dt = 0.0
        self.device = device
        self.cuda = bool(device and str(device).startswith("cuda"))

    def __enter__(self):
        self.start = self.time()
        return self

    def __exit__(self, type, value, traceback):  # noqa: A002
        self.dt = self.time() - self.start
        self.t += self.dt

    def __str__(self):
        return f"Elapsed time is {self.t} s"

    def time(self):
        if self.cuda:
            torch.cuda.synchronize(self.device)
        return time.time()


_NMS_WORK: "dict[tuple, torch.Tensor]" = {}
_NMS_WORK_MAX = 8


def _nms_workspace(device, batch, anchors, max_cand, ml_nc: int = 0) -> torch.Tensor:
    """Scratch of one NMS call, cached per (device, STREAM, shape): two callers on two streams never share candidate /
    sort / mask arrays, and a buffer is allocated, used and eventually freed on one stream only, which is what the
    caching allocator's reuse rule assumes.  At most _NMS_WORK_MAX buffers are kept (oldest dropped first)."""
    stream = torch.cuda.current_stream(device).cuda_stream
    key = (str(device), int(stream), batch, anchors, max_cand, ml_nc)
    w = _NMS_WORK.pop(key, None)
    if w is None:
        if ml_nc:
            nbytes = _lib.lib().yl_nms_ml_workspace_bytes(batch, ml_nc, anchors, max_cand)
        else:
            nbytes = _lib.lib().yl_nms_workspace_bytes(batch, anchors, max_cand)
        if nbytes < 0:
            raise RuntimeError("yl_nms_workspace_bytes failed")
        while len(_NMS_WORK) >= _NMS_WORK_MAX:
            _NMS_WORK.pop(next(iter(_NMS_WORK)))
        w = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    _NMS_WORK[key] = w  # most recently used last
    return w


def nms_padded(prediction: torch.Tensor, conf_thres: float, iou_thres: float, classes=None, agnostic=False,
               max_det: int = 300, max_nms: int = 30000, max_wh: float = 7680.0,
               scale: Optional[torch.Tensor] = None, multi_label: bool = False):
    """GPU NMS returning padded ``(out[B, max_det, 6], count[B])`` device tensors; no host sync.
    ``multi_label``: every (anchor, class) pair above the threshold is a candidate (validator mode)."""
    if prediction.device.type != "cuda":
        raise RuntimeError("non_max_suppression runs on the GPU only (HIP kernel, no CPU fallback)")
    if prediction.dtype != torch.float32 or not prediction.is_contiguous():
        prediction = prediction.float().contiguous()
    B, no, A = prediction.shape
    nc = no - 4
    dev = prediction.device
    multi_label = bool(multi_label) and nc > 1  # ops.py:208
    max_cand = min(int(max_nms), A * nc if multi_label else A)
    work = _nms_workspace(dev, B, A, max_cand, nc if multi_label else 0)
    out = torch.empty((B, max_det, 6), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    cls_t = None
    if classes is not None:
        cls_t = torch.as_tensor(list(np.atleast_1d(classes)), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    fn = _lib.lib().yl_nms_ml if multi_label else _lib.lib().yl_nms
    rc = fn(prediction.data_ptr(), B, nc, A, float(conf_thres), float(iou_thres), int(max_det),
            int(bool(agnostic)), cls_t.data_ptr() if cls_t is not None else None,
            int(cls_t.numel()) if cls_t is not None else 0, max_cand, float(max_wh),
            scale.data_ptr() if scale is not None else None, out.data_ptr(), cnt.data_ptr(),
            work.data_ptr(), stream)
    _lib.check(rc, "yl_nms")
    return out, cnt


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, labels=(), max_det=300, nc=0, max_time_img=0.05, max_nms=30000,
                        max_wh=7680, in_place=True, rotated=False) -> List[torch.Tensor]:
    """Reference ``ops.py:138-273``: (B, 4+nc, A) -> list of (n_i, 6) [x1, y1, x2, y2, conf, cls].

    Differences, all documented: the input is never modified (``in_place`` is accepted and ignored);
    there is no wall-clock bail-out (``max_time_img`` ignored; reference defect ops.py:269-271);
    ``multi_label`` (the validator's mode, ops.py:239-241) runs ``yl_nms_ml``; ``labels`` (autolabelling
    a-priori boxes) and ``rotated`` are not part of the detect path and raise.
    """
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if rotated or (labels is not None and len(labels)):
        raise NotImplementedError("labels / rotated NMS are outside the YOLO11 detect path")
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if prediction.shape[-1] == 6:
        raise NotImplementedError("end2end (B, N, 6) predictions are outside the YOLO11 predict path")
    out, cnt = nms_padded(prediction, conf_thres, iou_thres, classes, agnostic, max_det, max_nms, float(max_wh),
                          multi_label=multi_label)
    counts = cnt.tolist()  # the one host sync of post-processing
    return [out[i, :n] for i, n in enumerate(counts)]


def clip_boxes(boxes, shape):
    """Clip xyxy boxes to an (h, w) image in place (reference ops.py:276-295)."""
    if isinstance(boxes, torch.Tensor):
        boxes[..., 0] = boxes[..., 0].clamp(0, shape[1])
        boxes[..., 1] = boxes[..., 1].clamp(0, shape[0])
        boxes[..., 2] = boxes[..., 2].clamp(0, shape[1])
        boxes[..., 3] = boxes[..., 3].clamp(0, shape[0])
    else:
        boxes[..., [0, 2]] = boxes[..., [0, 2]].clip(0, shape[1])
        boxes[..., [1, 3]] = boxes[..., [1, 3]].clip(0, shape[0])
    return boxes


def scale_boxes_params(img1_shape, img0_shape):
    """(gain, (pad_w, pad_h)) of reference ops.py:80-88, with Python's banker's ``round``."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1),
           round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    return gain, pad


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True, xywh=False):
    """Rescale boxes from the letterboxed shape to the original image (reference ops.py:66-98)."""
    if ratio_pad is None:
        gain, pad = scale_boxes_params(img1_shape, img0_shape)
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    if padding:
        boxes[..., 0] -= pad[0]
        boxes[..., 1] -= pad[1]
        if not xywh:
            boxes[..., 2] -= pad[0]
            boxes[..., 3] -= pad[1]
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


def xywh2xyxy(x):
    """(cx, cy, w, h) -> (x1, y1, x2, y2) (reference ops.py:372-389)."""
    assert x.shape[-1] == 4, f"input shape last dimension expected 4 but input shape is {x.shape}"
    y = torch.empty_like(x) if isinstance(x, torch.Tensor) else np.empty_like(x)
    xy, wh = x[..., :2], x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def xyxy2xywh(x):
    """(x1, y1, x2, y2) -> (cx, cy, w, h) (reference ops.py:352-369)."""
    assert x.shape[-1] == 4, f"input shape last dimension expected 4 but input shape is {x.shape}"
    y = torch.empty_like(x) if isinstance(x, torch.Tensor) else np.empty_like(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def make_divisible(x, divisor):
    import math

    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def convert_torch2numpy_batch(batch: torch.Tensor) -> np.ndarray:
    """BCHW float [0,1] -> BHWC uint8 on the host (reference ops.py:487-488)."""
    return (batch.permute(0, 2, 3, 1).contiguous() * 255).clamp(0, 255).to(torch.uint8).cpu().numpy()
