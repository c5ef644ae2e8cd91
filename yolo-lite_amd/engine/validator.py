"""DetectionValidator: the second caller of the accelerated path (reference ``yololite/engine/validator.py``).

Same forward as the predictor (HIP engine), then the validator's own post-processing: NMS at conf 0.001 with
``multi_label=True`` (validator.py:281-291 -> ``yl_nms_ml``), matching of predictions to labels at ten IoU
thresholds (validator.py:192-229) and AP per class (utils/metrics.py).  Out of scope, as in DESIGN.md: dataset
construction (``check_det_dataset`` / ``build_yolo_dataset``: pass a ``dataloader`` yielding the reference's
batch dicts), validation inside training (``trainer=``), plots, confusion matrix, COCO/LVIS json evaluation.

A batch is the dict the reference's dataloader produces (data/dataset.py collate): ``img`` (B, 3, H, W) uint8,
``cls`` (n, 1), ``bboxes`` (n, 4) normalised xywh in the letterboxed image, ``batch_idx`` (n,), ``ori_shape``
[(h0, w0)] * B, ``ratio_pad`` [((gain_h, gain_w), (pad_w, pad_h))] * B, ``im_file`` [str] * B.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from ..cfg import get_cfg, get_save_dir
from ..nn.autobackend import AutoBackend
from ..utils import LOGGER, ops
from ..utils.checks import check_imgsz
from ..utils.metrics import DetMetrics, box_iou
from ..utils.torch_utils import select_device, smart_inference_mode


class DetectionValidator:
    def __init__(self, dataloader=None, save_dir=None, pbar=None, args=None):
        self.args = get_cfg(overrides=args)
        self.dataloader = dataloader
        self.pbar = pbar
        self.stride = None
        self.data = None
        self.device = None
        self.batch_i = None
        self.training = False
        self.names = None
        self.seen = None
        self.stats = None
        self.nc = None
        self.jdict = None
        self.speed = {"preprocess": 0.0, "inference": 0.0, "loss": 0.0, "postprocess": 0.0}
        self.nt_per_class = None
        self.nt_per_image = None
        self.class_map = None
        self.args.task = "detect"
        self.iouv = torch.linspace(0.5, 0.95, 10)  # mAP@0.5:0.95
        self.niou = self.iouv.numel()
        self._save_dir = save_dir
        if self.args.conf is None:
            self.args.conf = 0.001  # validator.py:86-87
        self.args.imgsz = check_imgsz(self.args.imgsz, max_dim=1)
        self.metrics = DetMetrics(save_dir=save_dir or Path("."))

    @property
    def save_dir(self) -> Path:
        if self._save_dir is None:
            self._save_dir = get_save_dir(self.args)
        return Path(self._save_dir)

    # ------------------------------------------------------------------ driver
    @smart_inference_mode()
    def __call__(self, trainer=None, model=None):
        if trainer is not None:
            raise NotImplementedError("validation inside training is outside the accelerated inference path")
        if self.dataloader is None:
            raise NotImplementedError("dataset construction is out of scope: pass dataloader= (an iterable of the "
                                      "reference's batch dicts) to DetectionValidator")
        if not isinstance(model, AutoBackend):
            model = AutoBackend(weights=model if model is not None else self.args.model,
                                device=select_device(self.args.device, self.args.batch, verbose=False),
                                data=self.args.data, fp16=self.args.half, fuse=True, verbose=False)
        self.device = model.device
        self.stride = model.stride
        model.eval()
        profilers = [ops.Profile(device=self.device) for _ in range(4)]
        self.init_metrics(model)
        n_img = 0
        for batch_i, batch in enumerate(self.dataloader):
            self.batch_i = batch_i
            with profilers[0]:
                batch = self.preprocess(batch)
            with profilers[1]:
                preds = model(batch["img"])
            with profilers[3]:
                preds = self.postprocess(preds)
            self.update_metrics(preds, batch)
            n_img += len(batch["img"])
        stats = self.get_stats()
        self.speed = dict(zip(self.speed.keys(), (p.t / max(n_img, 1) * 1e3 for p in profilers)))
        self.finalize_metrics()
        self.print_results()
        if self.args.save_json and self.jdict:
            self.save_dir.mkdir(parents=True, exist_ok=True)
            with open(self.save_dir / "predictions.json", "w") as f:
                json.dump(self.jdict, f)
        return stats

    # ------------------------------------------------------------------ stages
    def preprocess(self, batch):
        """Device move (validator.py:263-279).  The image stays uint8: the engine's first layer divides by 255,
        which is what the reference does here in fp32."""
        img = batch["img"].to(self.device, non_blocking=True)
        batch["img"] = img if img.dtype == torch.uint8 else img.float() / (255 if img.max() > 1 else 1)
        for k in ("batch_idx", "cls", "bboxes"):
            batch[k] = batch[k].to(self.device)
        return batch

    def postprocess(self, preds):
        return ops.non_max_suppression(preds, self.args.conf, self.args.iou, multi_label=True,
                                       agnostic=self.args.single_cls or self.args.agnostic_nms,
                                       max_det=self.args.max_det)

    def init_metrics(self, model):
        self.names = model.names
        self.nc = len(model.names)
        self.class_map = list(range(self.nc))
        self.metrics.names = self.names
        self.seen = 0
        self.jdict = []
        self.stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[], target_img=[])

    def _prepare_batch(self, si, batch):
        idx = batch["batch_idx"] == si
        cls = batch["cls"][idx].reshape(-1)
        bbox = batch["bboxes"][idx]
        ori_shape = tuple(int(v) for v in batch["ori_shape"][si])
        imgsz = tuple(batch["img"].shape[2:])
        ratio_pad = batch["ratio_pad"][si] if batch.get("ratio_pad") is not None else None
        if len(cls):
            whwh = torch.tensor([imgsz[1], imgsz[0], imgsz[1], imgsz[0]], device=bbox.device, dtype=bbox.dtype)
            bbox = ops.xywh2xyxy(bbox) * whwh
            ops.scale_boxes(imgsz, bbox, ori_shape, ratio_pad=ratio_pad)  # labels in the original image's pixels
        return {"cls": cls, "bbox": bbox, "ori_shape": ori_shape, "imgsz": imgsz, "ratio_pad": ratio_pad}

    def _prepare_pred(self, pred, pbatch):
        predn = pred.clone()
        ops.scale_boxes(pbatch["imgsz"], predn[:, :4], pbatch["ori_shape"], ratio_pad=pbatch["ratio_pad"])
        return predn

    def match_predictions(self, pred_classes, true_classes, iou):
        """(N, T) bool: is detection n a true positive at IoU threshold t.  Same outcome as the reference's matching
        (validator.py:192-229: per threshold, pairs of one class above the threshold sorted by IoU, one label per detection, then
        one detection per label) computed for all T thresholds at once from two facts about that procedure: a detection's partner is
        its best-overlapping label of its own class whatever the threshold (the threshold only decides whether the pair counts),
        and a label wanted by several detections goes to the one with the LOWEST index (the reference's second ``np.unique`` runs
        over rows already re-ordered by detection index) -- i.e. the most confident one, detections being score-sorted."""
        n, dev = pred_classes.shape[0], pred_classes.device
        T = self.niou
        correct = torch.zeros((n, T), dtype=torch.bool, device=dev)
        L = true_classes.shape[0]
        if n == 0 or L == 0:
            return correct
        same = iou * (true_classes[:, None] == pred_classes[None, :]).to(iou.dtype)   # (L, n): IoU where the classes agree, else 0
        best_iou, partner = same.max(0)                                                 # per detection
        counts = best_iou[:, None] >= self.iouv.to(device=dev, dtype=iou.dtype)[None, :]  # (n, T): the pair counts at threshold t
        det = torch.arange(n, device=dev)[:, None].expand(n, T)
        claim = torch.where(counts, det, torch.full_like(det, n))
        winner = torch.full((L, T), n, dtype=det.dtype, device=dev)
        winner.scatter_reduce_(0, partner[:, None].expand(n, T), claim, reduce="amin")  # lowest claiming detection per (label, t)
        return counts & (winner[partner] == det)

    def _process_batch(self, detections, gt_bboxes, gt_cls):
        return self.match_predictions(detections[:, 5], gt_cls, box_iou(gt_bboxes, detections[:, :4]))

    def update_metrics(self, preds, batch):
        """One statistics record per image (reference validator.py:313-360): confidences, predicted classes, the (n, T) true-positive
        table against the image's labels, the labels' classes.  An image with neither detections nor labels leaves no record."""
        empty = torch.zeros(0, device=self.device)
        for si, pred in enumerate(preds):
            self.seen += 1
            labels = self._prepare_batch(si, batch)
            gt_cls, gt_box = labels["cls"], labels["bbox"]
            n = len(pred)
            if n == 0 and len(gt_cls) == 0:
                continue
            if n and self.args.single_cls:
                pred[:, 5] = 0
            predn = self._prepare_pred(pred, labels) if n else pred
            tp = self._process_batch(predn, gt_box, gt_cls) if (n and len(gt_cls)) else torch.zeros(n, self.niou, dtype=torch.bool,
                                                                                                     device=self.device)
            record = dict(tp=tp, conf=predn[:, 4] if n else empty, pred_cls=predn[:, 5] if n else empty, target_cls=gt_cls,
                          target_img=gt_cls.unique())
            for k, v in record.items():
                self.stats[k].append(v)
            if n and self.args.save_json:
                self.pred_to_json(predn, batch["im_file"][si])
            if n and self.args.save_txt:
                self.save_one_txt(predn, self.args.save_conf, labels["ori_shape"],
                                  self.save_dir / "labels" / f'{Path(batch["im_file"][si]).stem}.txt')

    def finalize_metrics(self, *args, **kwargs):
        self.metrics.speed = self.speed

    def get_stats(self):
        stats = {k: (torch.cat(v, 0).cpu().numpy() if len(v) else np.zeros(0)) for k, v in self.stats.items()}
        self.nt_per_class = np.bincount(stats["target_cls"].astype(int), minlength=self.nc)
        self.nt_per_image = np.bincount(stats["target_img"].astype(int), minlength=self.nc)
        stats.pop("target_img", None)
        if len(stats) and stats["tp"].any():
            self.metrics.process(**stats)
        return self.metrics.results_dict

    def print_results(self):
        pf = "%22s" + "%11i" * 2 + "%11.3g" * len(self.metrics.keys)
        if self.args.verbose:
            LOGGER.info(self.get_desc())
        LOGGER.info(pf % ("all", self.seen, self.nt_per_class.sum(), *self.metrics.mean_results()))
        if self.nt_per_class.sum() == 0:
            LOGGER.warning("WARNING: no labels found in the validation set, cannot compute metrics")
        if self.args.verbose and self.nc > 1 and len(self.stats):
            for i, c in enumerate(self.metrics.ap_class_index):
                LOGGER.info(pf % (self.names[c], self.nt_per_image[c], self.nt_per_class[c], *self.metrics.class_result(i)))

    def get_desc(self):
        return ("%22s" + "%11s" * 6) % ("Class", "Images", "Instances", "Box(P", "R", "mAP50", "mAP50-95)")

    def save_one_txt(self, predn, save_conf, shape, file):
        from .results import Results

        Results(np.zeros((shape[0], shape[1]), dtype=np.uint8), path=None, names=self.names,
                boxes=predn[:, :6]).save_txt(file, save_conf=save_conf)

    def pred_to_json(self, predn, filename):
        """COCO-format rows (validator.py:462-478): top-left xywh, 3 / 5 decimals."""
        stem = Path(filename).stem
        image_id = int(stem) if stem.isnumeric() else stem
        box = ops.xyxy2xywh(predn[:, :4])
        box[:, :2] -= box[:, 2:] / 2
        for p, b in zip(predn.tolist(), box.tolist()):
            self.jdict.append({"image_id": image_id, "category_id": self.class_map[int(p[5])],
                               "bbox": [round(x, 3) for x in b], "score": round(p[4], 5)})
