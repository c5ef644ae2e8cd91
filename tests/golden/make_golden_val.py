"""Golden vectors for the VALIDATOR path, made by importing the REFERENCE (build container only):

    python tests/golden/make_golden_val.py        # writes tests/golden/golden_val_v1.npz

Same import-time stand-ins as make_golden.py (``cv2``; ``torchvision.ops.nms`` = our restatement of the published
greedy NMS, so these vectors pin the reference's own code AROUND it): ``ops.non_max_suppression(multi_label=True)``
(utils/ops.py:208, 239-241 -- the validator's postprocess), ``metrics.box_iou / smooth / compute_ap / ap_per_class``
(utils/metrics.py) and ``DetectionValidator.match_predictions`` (engine/validator.py:192-229).  Only numbers the
reference computed are stored.
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
spec = importlib.util.spec_from_file_location("make_golden", HERE / "make_golden.py")
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)  # installs the stand-ins and puts /root/reference on sys.path (main() is not run)

from yololite.engine.validator import DetectionValidator  # noqa: E402
from yololite.utils import metrics as ref_metrics  # noqa: E402
from yololite.utils import ops as ref_ops  # noqa: E402


def main():
    out = {}
    g = torch.Generator().manual_seed(77)
    # ---- multi-label NMS: dense class scores, a few hot anchors, overlapping boxes
    B, nc, A = 2, 6, 300
    p = torch.zeros(B, 4 + nc, A)
    p[:, 0:2] = torch.rand(B, 2, A, generator=g) * 300 + 20
    p[:, 2:4] = torch.rand(B, 2, A, generator=g) * 80 + 10
    sc = torch.rand(B, nc, A, generator=g) ** 6
    hot = torch.rand(B, 1, A, generator=g) < 0.15
    sc = torch.where(hot, torch.rand(B, nc, A, generator=g), sc)
    p[:, 4:] = sc
    out["ml_pred"] = p.numpy()
    cases = {"c001": dict(conf_thres=0.001, iou_thres=0.7), "c25": dict(conf_thres=0.25, iou_thres=0.45),
             "agn": dict(conf_thres=0.05, iou_thres=0.6, agnostic=True),
             "cap": dict(conf_thres=0.001, iou_thres=0.7, max_nms=500),
             "cls": dict(conf_thres=0.01, iou_thres=0.7, classes=[1, 4]), "md": dict(conf_thres=0.01, iou_thres=0.7, max_det=17)}
    for name, kw in cases.items():
        res = ref_ops.non_max_suppression(p.clone(), multi_label=True, in_place=False, max_time_img=1e9, **kw)
        out[f"ml_{name}_n"] = np.asarray([len(r) for r in res], dtype=np.int64)
        out[f"ml_{name}_out"] = np.concatenate([r.numpy() for r in res], 0)
        print("multi-label nms", name, out[f"ml_{name}_n"].tolist())
    # ---- metrics
    b1 = torch.rand(7, 4, generator=g) * 50
    b1[:, 2:] += b1[:, :2] + 5
    b2 = torch.rand(9, 4, generator=g) * 50
    b2[:, 2:] += b2[:, :2] + 5
    out["iou_b1"], out["iou_b2"] = b1.numpy(), b2.numpy()
    out["iou_out"] = ref_metrics.box_iou(b1, b2).numpy()
    y = np.random.default_rng(1).random(1000)
    out["smooth_in"], out["smooth_out"] = y, ref_metrics.smooth(y, 0.1)
    rng = np.random.default_rng(2)
    n, ncls = 400, 5
    conf = rng.random(n)
    pred_cls = rng.integers(0, ncls, n).astype(np.float64)
    tp = (rng.random((n, 10)) < np.linspace(0.8, 0.2, 10)[None, :] * conf[:, None] ** 0.3)
    tp = np.logical_and.accumulate(tp, axis=1)  # a TP at a high IoU threshold is one at the lower ones
    target_cls = np.concatenate([rng.integers(0, ncls - 1, 150), [ncls]]).astype(np.float64)  # class 4: no labels; class 5: no preds
    r = ref_metrics.ap_per_class(tp, conf, pred_cls, target_cls, plot=False, names={})
    out["apc_tp"], out["apc_conf"], out["apc_pred_cls"], out["apc_target_cls"] = tp, conf, pred_cls, target_cls
    for k, v in zip(("tp_n", "fp_n", "p", "r", "f1", "ap", "classes", "p_curve", "r_curve", "f1_curve", "px", "prec_values"), r):
        out["apc_" + k] = np.asarray(v)
    rec = np.sort(rng.random(30))
    prec = np.sort(rng.random(30))[::-1].copy()
    ap, mpre, mrec = ref_metrics.compute_ap(rec, prec)
    out["cap_rec"], out["cap_prec"], out["cap_ap"], out["cap_mpre"], out["cap_mrec"] = rec, prec, np.asarray([ap]), mpre, mrec
    # ---- validator matching
    holder = types.SimpleNamespace(iouv=torch.linspace(0.5, 0.95, 10))
    gt = torch.rand(12, 4, generator=g) * 100
    gt[:, 2:] += gt[:, :2] + 10
    det = gt[torch.randint(0, 12, (40,), generator=g)] + torch.randn(40, 4, generator=g) * 3
    gt_cls = torch.randint(0, 3, (12,), generator=g).float()
    det_cls = torch.randint(0, 3, (40,), generator=g).float()
    iou = ref_metrics.box_iou(gt, det)
    correct = DetectionValidator.match_predictions(holder, det_cls, gt_cls, iou)
    out["match_gt"], out["match_det"], out["match_gt_cls"], out["match_det_cls"] = gt.numpy(), det.numpy(), gt_cls.numpy(), det_cls.numpy()
    out["match_correct"] = correct.numpy()
    print("match_predictions true positives per threshold", correct.sum(0).tolist())
    np.savez_compressed(HERE / "golden_val_v1.npz", **out)
    print("wrote", HERE / "golden_val_v1.npz", sum(v.nbytes for v in out.values()), "bytes")


if __name__ == "__main__":
    main()
