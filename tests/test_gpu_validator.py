"""-m gpu: DetectionValidator end to end on the HIP path: forward, multi-label NMS at conf
0.001, label matching, AP.  Labels are made from the model's own confident single-label detections, so the
expected mAP50 is the metric's ceiling (0.995, see tests/test_metrics.py) and recall is 1."""
import json

import numpy as np
import pytest
import torch

from yololite_amd import YOLOLite
from yololite_amd.engine.validator import DetectionValidator
from yololite_amd.utils import ops

pytestmark = pytest.mark.gpu


def _make_loader(model, device, n_batches=2, B=4, H=192, W=256, h0=480, w0=640):
    """Batches in the reference's collate format with labels taken from predictions at conf 0.5."""
    g = torch.Generator().manual_seed(3)
    gain = min(H / h0, W / w0)
    pad = (round((W - w0 * gain) / 2 - 0.1), round((H - h0 * gain) / 2 - 0.1))
    loader = []
    for bi in range(n_batches):
        img = torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8)
        res = model.predict(img.to(device), conf=0.5, iou=0.7, device=str(device), save=False, verbose=False)  # the validator's NMS IoU
        cls, boxes, bidx = [], [], []
        for i, r in enumerate(res):
            d = r.boxes.data.cpu()  # pixels of the (H, W) tensor = the letterboxed image here
            if len(d):
                xywh = ops.xyxy2xywh(d[:, :4].clone()) / torch.tensor([W, H, W, H])
                boxes.append(xywh)
                cls.append(d[:, 5:6])
                bidx.append(torch.full((len(d),), float(i)))
        loader.append({"img": img, "cls": torch.cat(cls) if cls else torch.zeros(0, 1),
                       "bboxes": torch.cat(boxes) if boxes else torch.zeros(0, 4),
                       "batch_idx": torch.cat(bidx) if bidx else torch.zeros(0),
                       "ori_shape": [(h0, w0)] * B, "ratio_pad": [((gain, gain), pad)] * B,
                       "im_file": [f"{bi * B + i:012d}.jpg" for i in range(B)]})
    return loader


def test_validator_end_to_end(gpu_device, tmp_path):
    model = YOLOLite("yolo11n.yaml")
    loader = _make_loader(model, gpu_device)
    n_labels = sum(len(b["cls"]) for b in loader)
    assert n_labels > 10, "the synthetic model should produce confident detections to use as labels"
    v = DetectionValidator(dataloader=loader, save_dir=tmp_path,
                           args=dict(model="yolo11n.yaml", device=str(gpu_device), save_json=True, verbose=False))
    stats = v(model=model.model)
    assert v.seen == 8 and int(v.nt_per_class.sum()) == n_labels
    assert v.args.conf == 0.001 and set(v.speed) == {"preprocess", "inference", "loss", "postprocess"}
    assert stats["metrics/recall(B)"] > 0.99
    assert stats["metrics/mAP50(B)"] > 0.97       # ceiling 0.995
    assert stats["metrics/mAP50-95(B)"] > 0.95    # the labels ARE the model's boxes (up to the NMS cluster choice)
    rows = json.loads((tmp_path / "predictions.json").read_text())
    assert len(rows) > n_labels and set(rows[0]) == {"image_id", "category_id", "bbox", "score"}
    assert rows[0]["image_id"] == 0 and len(rows[0]["bbox"]) == 4
    # YOLOLite.val is the same thing behind the reference's entry point (model.py:101-107)
    m = model.val(dataloader=loader, device=str(gpu_device), verbose=False, project=str(tmp_path), name="v")
    assert abs(m.results_dict["metrics/mAP50(B)"] - stats["metrics/mAP50(B)"]) < 1e-12


def test_validator_shifted_labels_lower_the_map(gpu_device):
    """Shifting every label by 15 % of its width (IoU with its prediction ~0.74) must cost far more AP at the
    high IoU thresholds than at 0.5."""
    model = YOLOLite("yolo11n.yaml")
    loader = _make_loader(model, gpu_device, n_batches=1)
    args = dict(model="yolo11n.yaml", device=str(gpu_device), verbose=False)
    base = DetectionValidator(dataloader=loader, args=args)(model=model.model)
    for b in loader:
        b["bboxes"] = b["bboxes"].clone()
        b["bboxes"][:, 0] += 0.15 * b["bboxes"][:, 2]
    moved = DetectionValidator(dataloader=loader, args=args)(model=model.model)
    assert base["metrics/mAP50-95(B)"] > 0.95
    assert moved["metrics/mAP50-95(B)"] < base["metrics/mAP50-95(B)"] - 0.3
    assert moved["metrics/mAP50(B)"] > moved["metrics/mAP50-95(B)"] + 0.15
