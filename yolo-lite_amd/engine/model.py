"""``YOLOLite`` / ``YOLO`` facade (reference ``yololite/engine/model.py:17-146``).

``YOLOLite("yolo11n.yaml")`` builds from the architecture table, anything else loads a ``.pt``
checkpoint; ``model(source)`` == ``model.predict(source)`` returns ``list[Results]``.
Training is outside the accelerated path and raises ``NotImplementedError``; ``val`` runs the
DetectionValidator on a caller-supplied dataloader.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Union

from ..cfg import DEFAULT_CFG_DICT
from ..nn.tasks import DetectionModel, attempt_load_one_weight, yaml_model_load
from .predictor import DetectionPredictor
from .results import Results


class YOLOLite:
    def __init__(self, model: Union[str, Path] = "yolo11n.pt", task: str = None, verbose: bool = False) -> None:
        self.callbacks = None
        self.predictor = None
        self.model = None
        self.trainer = None
        self.ckpt = None
        self.cfg = None
        self.ckpt_path = None
        self.overrides = {}
        self.metrics = None
        self.task = task
        model = str(model).strip()
        if Path(model).suffix in {".yaml", ".yml"}:
            self._new(model, task=task, verbose=verbose)
        else:
            self._load(model)

    def __call__(self, source=None, stream: bool = False, **kwargs) -> list:
        return self.predict(source, stream, **kwargs)

    def _new(self, cfg: str, task=None, verbose=False) -> None:
        cfg_dict = yaml_model_load(cfg)
        self.cfg = cfg
        self.task = task or "detect"
        self.model = DetectionModel(cfg_dict, verbose=verbose)
        self.overrides["model"] = self.cfg
        self.overrides["task"] = self.task
        self.model.args = {**DEFAULT_CFG_DICT, **self.overrides}
        self.model.task = self.task
        self.model_name = cfg

    def _load(self, weights: str) -> None:
        self.model, self.ckpt = attempt_load_one_weight(weights)
        self.task = self.model.args.get("task", "detect")
        self.overrides = self.model.args = {k: v for k, v in self.model.args.items()
                                            if k in {"imgsz", "data", "task", "single_cls"}}
        self.ckpt_path = self.model.pt_path
        self.overrides["model"] = weights
        self.overrides["task"] = self.task
        self.model_name = weights

    def predict(self, source=None, stream: bool = False, **kwargs) -> List[Results]:
        """Same argument merge as reference model.py:95-98 (conf 0.25, batch 1, save True, mode predict).

        Unlike the reference, the predictor (and its engines / captured graphs) is reused across calls
        when the arguments are unchanged; the reference rebuilds it every call (model.py:97-98).
        """
        custom = {"conf": 0.25, "batch": 1, "save": True, "mode": "predict"}
        keep = kwargs.pop("keep_orig_img", False)
        args = {**self.overrides, **custom, **kwargs}
        if self.predictor is None or getattr(self, "_pred_args", None) != (args, keep):
            backend = self.predictor.model if self.predictor is not None else None
            self.predictor = DetectionPredictor(overrides=args, keep_orig_img=keep)
            if backend is not None and backend.model is self.model and str(args.get("device")) == str(
                    (getattr(self, "_pred_args", None) or ({},))[0].get("device")):
                self.predictor.model = backend  # keep built engines
                self.predictor.device = backend.device
                self.predictor.args.half = backend.fp16
            else:
                self.predictor.setup_model(model=self.model)
            self._pred_args = (dict(args), keep)
        return self.predictor(source=source, stream=stream)

    def val(self, dataloader=None, **kwargs):
        """Reference model.py:101-107 with one addition: ``dataloader=`` (an iterable of the reference's batch
        dicts), because dataset construction is out of scope here (see engine/validator.py)."""
        from .validator import DetectionValidator

        args = {**self.overrides, "rect": True, **kwargs, "mode": "val"}
        validator = DetectionValidator(dataloader=dataloader, args=args)
        validator(model=self.model)
        self.metrics = validator.metrics
        return validator.metrics

    def export(self, format: str = "safetensors", file=None, folded: bool = True) -> str:
        """Write the model as a flat table of tensors (``nn/export.py``): BN-folded filters and biases plus the
        architecture, names and strides in the header; ``YOLOLite("x.safetensors")`` reads it back without
        unpickling anything.  Other formats (onnx, engine ...) are not part of this path."""
        if format != "safetensors":
            raise NotImplementedError(f"export format '{format}' is outside the MI355X predict path (only 'safetensors')")
        from ..nn.export import save_safetensors

        if file is None:
            file = Path(str(self.ckpt_path or self.model_name or "model")).with_suffix(".safetensors")
        return save_safetensors(self.model, file, folded=folded)

    def train(self, **kwargs):
        raise NotImplementedError("training is not part of the MI355X predict path")

    def info(self):
        return self.model.info()

    @property
    def names(self):
        return self.model.names


YOLO = YOLOLite  # Ultralytics-style alias
