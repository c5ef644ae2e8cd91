"""Result containers with the reference's accessors (``yololite/engine/results.py:13-101, 443-580``).

``save_txt``, ``summary``, ``to_json`` / ``tojson`` and ``verbose`` are plain Python; ``plot`` / ``save`` /
``save_crop`` are result sinks outside the accelerated path, drawn with PIL (utils/plotting.py) because the
reference's default cv2 drawing is unavailable.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from ..utils import SimpleClass, ops


class BaseTensor(SimpleClass):
    def __init__(self, data, orig_shape) -> None:
        assert isinstance(data, (torch.Tensor, np.ndarray)), "data must be torch.Tensor or np.ndarray"
        self.data = data
        self.orig_shape = orig_shape

    @property
    def shape(self):
        return self.data.shape

    def cpu(self):
        return self if isinstance(self.data, np.ndarray) else self.__class__(self.data.cpu(), self.orig_shape)

    def numpy(self):
        return self if isinstance(self.data, np.ndarray) else self.__class__(self.data.cpu().numpy(), self.orig_shape)

    def cuda(self):
        return self.__class__(torch.as_tensor(self.data).cuda(), self.orig_shape)

    def to(self, *args, **kwargs):
        return self.__class__(torch.as_tensor(self.data).to(*args, **kwargs), self.orig_shape)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.__class__(self.data[idx], self.orig_shape)


class Boxes(BaseTensor):
    """(n, 6) detections [x1, y1, x2, y2, conf, cls] in original-image pixels."""

    def __init__(self, boxes, orig_shape) -> None:
        if boxes.ndim == 1:
            boxes = boxes[None, :]
        n = boxes.shape[-1]
        assert n in {6, 7}, f"expected 6 or 7 values but got {n}"
        super().__init__(boxes, orig_shape)
        self.orig_shape = orig_shape

    @classmethod
    def from_record(cls, record, count: int, orig_shape) -> "Boxes":
        """The first ``count`` rows of a padded (max_det, 6) record of the fused NMS (``HipEngine.detect``), sliced on first
        use: the predictor builds one Results per image per call, and 32 tensor slices per batch cost more host time than
        launching the network (the reference slices eagerly, predictor.py:110-127; the values are the same)."""
        self = cls.__new__(cls)
        self.__dict__["_record"] = (record, int(count))
        self.orig_shape = orig_shape
        return self

    def __getattr__(self, name):  # only reached while ``data`` has not been materialised yet
        if name == "data":
            rec = self.__dict__.get("_record")
            if rec is not None:
                data = rec[0][: rec[1]]
                self.__dict__["data"] = data
                return data
        raise AttributeError(name)

    def __len__(self):
        rec = self.__dict__.get("_record")
        return rec[1] if rec is not None and "data" not in self.__dict__ else len(self.data)

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, -2]

    @property
    def cls(self):
        return self.data[:, -1]

    @property
    def id(self):
        return None

    @property
    def xywh(self):
        return ops.xyxy2xywh(self.xyxy)

    @property
    def xyxyn(self):
        xyxy = self.xyxy.clone() if isinstance(self.xyxy, torch.Tensor) else np.copy(self.xyxy)
        xyxy[..., [0, 2]] /= self.orig_shape[1]
        xyxy[..., [1, 3]] /= self.orig_shape[0]
        return xyxy

    @property
    def xywhn(self):
        xywh = ops.xyxy2xywh(self.xyxy)
        xywh[..., [0, 2]] /= self.orig_shape[1]
        xywh[..., [1, 3]] /= self.orig_shape[0]
        return xywh


class Results(SimpleClass):
    """Per-image inference result: ``boxes``, ``names``, ``path``, ``orig_img``/``orig_shape``, ``speed``.

    ``orig_img`` may be ``None`` on the tensor fast path (the reference rebuilds a uint8 copy of the
    input on the host, predictor.py:120-121, which costs a full D2H transfer; pass
    ``keep_orig_img=True`` to the predictor to get it); ``orig_shape`` is always set.
    """

    def __init__(self, orig_img, path, names, boxes=None, speed=None, orig_shape=None) -> None:
        self.orig_img = orig_img
        self.orig_shape = tuple(orig_img.shape[:2]) if orig_img is not None else tuple(orig_shape)
        self.boxes = boxes if isinstance(boxes, Boxes) else (Boxes(boxes, self.orig_shape) if boxes is not None else None)
        self.speed = speed if speed is not None else {"preprocess": None, "inference": None, "postprocess": None}
        self.names = names
        self.path = path
        self.save_dir = None
        self._keys = ("boxes",)

    def __getitem__(self, idx):
        return self._apply("__getitem__", idx)

    def __len__(self):
        for k in self._keys:
            v = getattr(self, k)
            if v is not None:
                return len(v)
        return 0

    def update(self, boxes=None):
        if boxes is not None:
            self.boxes = Boxes(ops.clip_boxes(boxes, self.orig_shape), self.orig_shape)

    def _apply(self, fn, *args, **kwargs):
        r = self.new()
        for k in self._keys:
            v = getattr(self, k)
            if v is not None:
                setattr(r, k, getattr(v, fn)(*args, **kwargs))
        return r

    def cpu(self):
        return self._apply("cpu")

    def numpy(self):
        return self._apply("numpy")

    def cuda(self):
        return self._apply("cuda")

    def to(self, *args, **kwargs):
        return self._apply("to", *args, **kwargs)

    def new(self):
        return Results(orig_img=self.orig_img, path=self.path, names=self.names, speed=self.speed,
                       orig_shape=self.orig_shape)

    def verbose(self) -> str:
        """'3 persons, 1 boat, ' style summary (reference results.py:246-263)."""
        s = ""
        boxes = self.boxes
        if boxes is not None and len(boxes):
            cls = boxes.cls if isinstance(boxes.cls, torch.Tensor) else torch.as_tensor(boxes.cls)
            for c in cls.unique():
                n = int((cls == c).sum())
                s += f"{n} {self.names[int(c)]}{'s' * (n > 1)}, "
        return s

    def summary(self, normalize=False, decimals=5):
        """List of dicts {name, class, confidence, box{x1,y1,x2,y2}} (reference results.py:319-352)."""
        out = []
        if self.boxes is None:
            return out
        data = self.boxes.data.cpu().numpy() if isinstance(self.boxes.data, torch.Tensor) else self.boxes.data
        h, w = self.orig_shape if normalize else (1, 1)
        for row in data:
            cid = int(row[5])
            out.append({"name": self.names[cid], "class": cid, "confidence": round(float(row[4]), decimals),
                        "box": {"x1": round(float(row[0]) / w, decimals), "y1": round(float(row[1]) / h, decimals),
                                "x2": round(float(row[2]) / w, decimals), "y2": round(float(row[3]) / h, decimals)}})
        return out

    def to_json(self, normalize=False, decimals=5):
        return json.dumps(self.summary(normalize=normalize, decimals=decimals), indent=2)

    def tojson(self, normalize=False, decimals=5):
        """Deprecated spelling of ``to_json`` (reference results.py:415-418 warns the same way)."""
        from ..utils import LOGGER

        LOGGER.warning("WARNING 'result.tojson()' is deprecated, replace with 'result.to_json()'.")
        return self.to_json(normalize, decimals)

    def to_df(self, normalize=False, decimals=5):
        """``summary()`` as a pandas DataFrame: columns name, class, confidence, box (reference results.py:354-371)."""
        import pandas as pd

        return pd.DataFrame(self.summary(normalize=normalize, decimals=decimals))

    def to_csv(self, normalize=False, decimals=5, *args, **kwargs):
        """CSV text of ``to_df`` (extra arguments go to ``DataFrame.to_csv``; reference results.py:373-392)."""
        return self.to_df(normalize=normalize, decimals=decimals).to_csv(*args, **kwargs)

    def to_xml(self, normalize=False, decimals=5, *args, **kwargs):
        """XML text of ``to_df`` (extra arguments go to ``DataFrame.to_xml``; an empty result gives the reference's empty document,
        results.py:394-413).  The reference insists on lxml; without it pandas' built-in ``etree`` writer produces the same document."""
        df = self.to_df(normalize=normalize, decimals=decimals)
        if df.empty:
            return '<?xml version="1.0" encoding="utf-8"?>\n<root></root>'
        if "parser" not in kwargs:
            try:
                import lxml  # noqa: F401
            except ImportError:
                kwargs["parser"] = "etree"
        return df.to_xml(*args, **kwargs)

    def save_txt(self, txt_file, save_conf=False):
        """Append 'cls cx cy w h [conf]' lines, normalised (reference results.py:265-290)."""
        boxes = self.boxes
        texts = []
        if boxes is not None:
            for d in boxes:
                c, conf = int(d.cls), float(d.conf)
                line = (c, *(float(v) for v in d.xywhn.reshape(-1)))
                line += (conf,) * save_conf
                texts.append(("%g " * len(line)).rstrip() % line)
        if texts:
            Path(txt_file).parent.mkdir(parents=True, exist_ok=True)
            with open(txt_file, "a") as f:
                f.writelines(t + "\n" for t in texts)

    def plot(self, conf=True, line_width=None, font_size=None, font="Arial.ttf", pil=False, img=None, labels=True,
             boxes=True, show=False, save=False, filename=None, color_mode="class"):
        """Annotated copy of the image (reference results.py:130-244).  Drawn with PIL (cv2 is absent): returns a
        BGR ``np.ndarray`` like the reference, or the PIL image when ``pil=True``."""
        from ..utils.plotting import Annotator, colors

        assert color_mode in {"instance", "class"}, f"Expected color_mode='instance' or 'class', not {color_mode}."
        if img is None:
            img = self.orig_img
            if isinstance(img, torch.Tensor):  # (1, 3, H, W) RGB in [0, 1] -> HWC BGR uint8
                img = (img[0].detach().permute(1, 2, 0).contiguous() * 255).to(torch.uint8).cpu().numpy()[..., ::-1]
        if img is None:
            raise ValueError("plot() needs the source image: predict with keep_orig_img=True or pass img=")
        ann = Annotator(np.ascontiguousarray(np.asarray(img)[..., ::-1]), line_width, font_size)  # BGR -> RGB
        if self.boxes is not None and boxes:
            for i, d in enumerate(reversed(self.boxes)):
                c = int(d.cls)
                name = self.names[c]
                label = (f"{name} {float(d.conf):.2f}" if conf else name) if labels else None
                ann.box_label(d.xyxy.reshape(-1), label, color=colors(c if color_mode == "class" else i))
        if show:  # (reference results.py:205-206 annotator.show(): cv2.imshow there, the platform's image viewer through PIL here)
            ann.im.show(title=str(self.path or "image0"))
        if save:
            ann.im.save(filename or f"results_{Path(self.path or 'image0').stem}.jpg")
        return ann.im if pil else ann.result()[..., ::-1].copy()

    def show(self, *args, **kwargs):
        """Display the annotated image (reference results.py:215-226: ``plot(show=True, ...)``)."""
        self.plot(show=True, *args, **kwargs)

    def save(self, filename=None, *args, **kwargs):
        """Save the annotated image (reference results.py: save)."""
        filename = filename or f"results_{Path(self.path or 'image0').name}"
        self.plot(save=True, filename=str(Path(filename).with_suffix(".jpg")), *args, **kwargs)
        return filename

    def save_crop(self, save_dir, file_name=Path("im.jpg")):
        """One crop per detection under save_dir/<class name>/ (reference results.py:289-312)."""
        from ..utils.plotting import save_one_box

        if self.orig_img is None:
            raise ValueError("save_crop() needs the source image: predict with keep_orig_img=True")
        for d in self.boxes:
            save_one_box(d.xyxy, np.asarray(self.orig_img).copy(),
                         file=Path(save_dir) / self.names[int(d.cls)] / Path(file_name).with_suffix(".jpg"), BGR=True)
