"""CPU: the validator-path restatements against vectors the REFERENCE computed (tests/golden/golden_val_v1.npz,
made by tests/golden/make_golden_val.py): multi-label NMS of the oracle, metrics, label matching."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import yolo11_oracle as oracle
from yololite_amd.engine.validator import DetectionValidator
from yololite_amd.utils import metrics

G = np.load(Path(__file__).parent / "golden" / "golden_val_v1.npz")
ML_CASES = {"c001": dict(conf_thres=0.001, iou_thres=0.7), "c25": dict(conf_thres=0.25, iou_thres=0.45),
            "agn": dict(conf_thres=0.05, iou_thres=0.6, agnostic=True),
            "cap": dict(conf_thres=0.001, iou_thres=0.7, max_nms=500),
            "cls": dict(conf_thres=0.01, iou_thres=0.7, classes=[1, 4]), "md": dict(conf_thres=0.01, iou_thres=0.7, max_det=17)}


def _split(name):
    n = G[f"ml_{name}_n"]
    out = G[f"ml_{name}_out"]
    return [out[int(n[:i].sum()): int(n[: i + 1].sum())] for i in range(len(n))]


@pytest.mark.parametrize("name", sorted(ML_CASES))
def test_oracle_multi_label_nms_matches_reference(name):
    res = oracle.non_max_suppression(torch.from_numpy(G["ml_pred"]), multi_label=True, **ML_CASES[name])
    for r, ref in zip(res, _split(name)):
        np.testing.assert_array_equal(r.numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ML_CASES))
def test_hip_multi_label_nms_matches_reference(gpu_device, name):
    from yololite_amd.utils import ops as hops

    res = hops.non_max_suppression(torch.from_numpy(G["ml_pred"]).to(gpu_device), multi_label=True, **ML_CASES[name])
    for r, ref in zip(res, _split(name)):
        np.testing.assert_array_equal(r.cpu().numpy(), ref)


def test_metrics_match_reference():
    iou = metrics.box_iou(torch.from_numpy(G["iou_b1"]), torch.from_numpy(G["iou_b2"])).numpy()
    np.testing.assert_allclose(iou, G["iou_out"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(metrics.smooth(G["smooth_in"], 0.1), G["smooth_out"], rtol=0, atol=1e-12)
    ap, mpre, mrec = metrics.compute_ap(G["cap_rec"], G["cap_prec"])
    assert abs(ap - float(G["cap_ap"][0])) < 1e-12
    np.testing.assert_array_equal(mpre, G["cap_mpre"])
    np.testing.assert_array_equal(mrec, G["cap_mrec"])
    r = metrics.ap_per_class(G["apc_tp"], G["apc_conf"], G["apc_pred_cls"], G["apc_target_cls"])
    for k, v in zip(("tp_n", "fp_n", "p", "r", "f1", "ap", "classes", "p_curve", "r_curve", "f1_curve", "px", "prec_values"), r):
        np.testing.assert_allclose(np.asarray(v, dtype=np.float64), np.asarray(G["apc_" + k], dtype=np.float64),
                                   rtol=0, atol=1e-12, err_msg=k)


def test_match_predictions_matches_reference():
    v = DetectionValidator(dataloader=[], args=dict(model="yolo11n.yaml"))
    gt, det = torch.from_numpy(G["match_gt"]), torch.from_numpy(G["match_det"])
    d6 = torch.cat([det, torch.ones(len(det), 1), torch.from_numpy(G["match_det_cls"])[:, None]], 1)
    got = v._process_batch(d6, gt, torch.from_numpy(G["match_gt_cls"]))
    np.testing.assert_array_equal(got.numpy(), G["match_correct"])
