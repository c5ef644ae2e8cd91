"""CPU: detection metrics and the validator's matching logic (host side of SURVEY 8(f)2)."""
import numpy as np
import pytest
import torch

from yololite_amd.engine.validator import DetectionValidator
from yololite_amd.utils.metrics import DetMetrics, ap_per_class, box_iou, compute_ap, smooth


def test_box_iou_known_values():
    a = torch.tensor([[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 4.0, 4.0]])
    b = torch.tensor([[0.0, 0.0, 10.0, 10.0], [5.0, 5.0, 15.0, 15.0], [20.0, 20.0, 30.0, 30.0]])
    iou = box_iou(a, b)
    assert iou.shape == (2, 3)
    np.testing.assert_allclose(iou[0].numpy(), [1.0, 25.0 / 175.0, 0.0], atol=1e-6)
    np.testing.assert_allclose(iou[1].numpy(), [0.16, 0.0, 0.0], atol=1e-6)


def test_compute_ap_perfect_and_half():
    # a perfect detector scores 0.995, not 1: the appended sentinel (recall 1, precision 0) duplicates x = 1 in
    # np.interp's table and the 101st sample reads 0 -- the reference's arithmetic (metrics.py:459-470), kept as is
    ap, _, _ = compute_ap(np.array([0.5, 1.0]), np.array([1.0, 1.0]))
    assert abs(ap - 0.995) < 1e-9
    # precision 1 up to recall 0.5, then nothing more is found: 1 on [0, 0.5], then a straight line down to the
    # sentinel (1, 0): 0.5 + 0.25
    ap, mpre, mrec = compute_ap(np.array([0.25, 0.5]), np.array([1.0, 1.0]))
    assert abs(ap - 0.75) < 1e-9 and mrec[0] == 0.0 and mrec[-1] == 1.0 and mpre[-1] == 0.0


def test_smooth_is_box_filter():
    y = np.arange(20, dtype=float)
    s = smooth(y, 0.1)
    assert s.shape == y.shape and abs(s[10] - 10.0) < 1e-9 and s[0] > 0  # interior unchanged for a ramp


def test_ap_per_class_two_classes():
    # class 0: 2 labels, detections (conf .9 TP, .8 FP, .7 TP); class 1: 1 label, one FP
    tp = np.zeros((4, 10), bool)
    tp[0] = True
    tp[2, :5] = True  # the third detection matches only up to IoU 0.7
    conf = np.array([0.9, 0.8, 0.7, 0.6])
    pred_cls = np.array([0, 0, 0, 1])
    target_cls = np.array([0, 0, 1])
    tp_n, fp_n, p, r, f1, ap, classes, *_ = ap_per_class(tp, conf, pred_cls, target_cls)
    assert list(classes) == [0, 1] and ap.shape == (2, 10)
    # class 0 @0.5: PR points (r .5, p 1), (r .5, p .5), (r 1, p 2/3): envelope 1 up to .5 then 2/3
    assert abs(ap[0, 0] - (0.5 * 1.0 + 0.5 * 2 / 3)) < 0.01
    # at IoU .95 only the first detection counts: precision 1 up to recall .5, the envelope then falls from 1/3
    # (three detections, one hit) to the sentinel
    assert abs(ap[0, 9] - (0.5 + 0.5 * (1 / 3) / 2)) < 0.01
    assert ap[1].max() == 0.0
    m = DetMetrics(names={0: "a", 1: "b"})
    m.process(tp, conf, pred_cls, target_cls)
    d = m.results_dict
    assert set(d) == {"metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)", "fitness"}
    assert abs(d["metrics/mAP50(B)"] - ap[:, 0].mean()) < 1e-12
    assert abs(d["fitness"] - (0.1 * ap[:, 0].mean() + 0.9 * ap.mean())) < 1e-12
    assert m.maps.shape == (2,)


def test_match_predictions_one_to_one():
    v = DetectionValidator(dataloader=[], args=dict(model="yolo11n.yaml"))
    gt = torch.tensor([[0.0, 0.0, 10.0, 10.0], [20.0, 20.0, 30.0, 30.0]])
    gt_cls = torch.tensor([1.0, 2.0])
    det = torch.tensor([[0.0, 0.0, 10.0, 10.0, 0.9, 1.0],    # exact match of label 0
                        [0.0, 0.0, 10.0, 9.0, 0.8, 1.0],     # second claim on label 0: loses (IoU 0.9)
                        [20.0, 20.0, 30.0, 30.0, 0.7, 1.0],  # right place, wrong class
                        [21.0, 20.0, 30.0, 30.0, 0.6, 2.0]])  # IoU 0.9 with label 1
    tp = v._process_batch(det, gt, gt_cls)
    assert tp.shape == (4, 10)
    assert tp[0].all() and not tp[1].any() and not tp[2].any()
    assert tp[3, :9].all() and not tp[3, 9]  # 0.9 >= thresholds .5 ... .9, not .95
    assert v.args.conf == 0.001


def test_validator_needs_a_dataloader():
    with pytest.raises(NotImplementedError):
        DetectionValidator(args=dict(model="yolo11n.yaml"))(model=None)
