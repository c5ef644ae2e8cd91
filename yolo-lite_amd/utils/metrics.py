"""Detection metrics for the validator (reference ``yololite/utils/metrics.py``: ``box_iou`` :51-70,
``smooth`` :387-393, ``compute_ap`` :445-474, ``ap_per_class`` :477-564, ``Metric`` :567-736,
``DetMetrics`` :739-837).  Host-side numpy; plotting of the curves is a result sink and is not provided."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import SimpleClass


def box_iou(box1: torch.Tensor, box2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """Pairwise IoU of (N, 4) and (M, 4) xyxy boxes -> (N, M) (reference metrics.py:51-70)."""
    a = box1.float()[:, None, :]
    b = box2.float()[None, :, :]
    wh = (torch.minimum(a[..., 2:], b[..., 2:]) - torch.maximum(a[..., :2], b[..., :2])).clamp_(0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    return inter / (area_a + area_b - inter + eps)


def smooth(y: np.ndarray, f: float = 0.05) -> np.ndarray:
    """Box filter over a fraction f of the curve, edges padded with the end values (metrics.py:387-393)."""
    nf = round(len(y) * f * 2) // 2 + 1  # odd
    pad = np.ones(nf // 2)
    yp = np.concatenate((pad * y[0], y, pad * y[-1]))
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def compute_ap(recall, precision):
    """AP = area under the precision envelope, 101-point interpolation as COCO (metrics.py:445-474)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    ap = float(np.sum((y[1:] + y[:-1]) * np.diff(x)) / 2.0)  # trapezoid rule (np.trapz)
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """Per-class precision / recall / F1 at the max-F1 confidence and AP at each IoU threshold
    (metrics.py:477-564; no plotting).  tp (n, niou) bool, conf (n,), pred_cls (n,), target_cls (m,)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, n_labels = np.unique(target_cls, return_counts=True)
    nc = classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap = np.zeros((nc, tp.shape[1]))
    p_curve, r_curve = np.zeros((nc, 1000)), np.zeros((nc, 1000))
    prec_values = []
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = n_labels[ci], int(sel.sum())
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)
        precision = tpc / (tpc + fpc)
        r_curve[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)
        p_curve[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], mpre, mrec = compute_ap(recall[:, j], precision[:, j])
            if j == 0:
                prec_values.append(np.interp(px, mrec, mpre))
    prec_values = np.array(prec_values)
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    best = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, best], r_curve[:, best], f1_curve[:, best]
    tp_n = (r * n_labels).round()
    fp_n = (tp_n / (p + eps) - tp_n).round()
    return tp_n, fp_n, p, r, f1, ap, classes.astype(int), p_curve, r_curve, f1_curve, px, prec_values


class Metric(SimpleClass):
    """Per-class detection results and their means (API of reference metrics.py:567-736): ``p r f1`` (nc,),
    ``all_ap`` (nc, 10) over IoU 0.50:0.95, ``ap_class_index``; everything else is derived from those."""

    _FITNESS_W = (0.0, 0.0, 0.1, 0.9)  # weights of [P, R, mAP50, mAP50-95]

    def __init__(self) -> None:
        self.p = self.r = self.f1 = self.all_ap = self.ap_class_index = ()
        self.nc = 0

    def _have(self) -> bool:
        return len(self.all_ap) > 0

    def _ap_at(self, j):
        """AP column j (IoU 0.5 + 0.05 j) per class."""
        return self.all_ap[:, j] if self._have() else []

    ap50 = property(lambda self: self._ap_at(0))
    ap = property(lambda self: self.all_ap.mean(1) if self._have() else [])
    mp = property(lambda self: float(np.mean(self.p)) if len(self.p) else 0.0)
    mr = property(lambda self: float(np.mean(self.r)) if len(self.r) else 0.0)
    map50 = property(lambda self: float(self._ap_at(0).mean()) if self._have() else 0.0)
    map75 = property(lambda self: float(self._ap_at(5).mean()) if self._have() else 0.0)
    map = property(lambda self: float(self.all_ap.mean()) if self._have() else 0.0)

    def mean_results(self):
        return [self.mp, self.mr, self.map50, self.map]

    def class_result(self, i):
        return self.p[i], self.r[i], self.ap50[i], self.ap[i]

    @property
    def maps(self):
        """mAP50-95 per class index; classes without data get the overall mean."""
        out = np.full(self.nc, self.map, dtype=np.float64)
        ap = self.ap
        for row, cls in enumerate(self.ap_class_index):
            out[cls] = ap[row]
        return out

    def fitness(self):
        return float(np.dot(self.mean_results(), self._FITNESS_W))

    def update(self, results):
        """results = ap_per_class(...)[2:]"""
        names = ("p", "r", "f1", "all_ap", "ap_class_index", "p_curve", "r_curve", "f1_curve", "px", "prec_values")
        for k, v in zip(names, results):
            setattr(self, k, v)


class DetMetrics(SimpleClass):
    """Front of the validator's metrics (API of reference metrics.py:739-837): ``process`` the accumulated
    statistics, then read ``results_dict`` / ``mean_results`` / ``maps`` / ``fitness``."""

    keys = ["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)"]

    def __init__(self, save_dir=Path("."), plot=False, on_plot=None, names=None) -> None:
        self.save_dir, self.plot, self.on_plot = save_dir, plot, on_plot
        self.names = names or {}
        self.box = Metric()
        self.speed = dict.fromkeys(("preprocess", "inference", "loss", "postprocess"), 0.0)
        self.task = "detect"

    def process(self, tp, conf, pred_cls, target_cls):
        self.box.nc = len(self.names)
        self.box.update(ap_per_class(tp, conf, pred_cls, target_cls)[2:])

    def mean_results(self):
        return self.box.mean_results()

    def class_result(self, i):
        return self.box.class_result(i)

    maps = property(lambda self: self.box.maps)
    fitness = property(lambda self: self.box.fitness())
    ap_class_index = property(lambda self: self.box.ap_class_index)

    @property
    def results_dict(self):
        return dict(zip([*self.keys, "fitness"], [*self.mean_results(), self.fitness]))
