"""Configuration: defaults, override merging and type checks with the reference's key set and
error behaviour (``yololite/cfg/default.yaml``, ``yololite/cfg/__init__.py:125-249``).

Only the keys below exist; unknown override keys raise ``SyntaxError`` exactly like the reference's
``check_dict_alignment`` (``cfg/__init__.py:219-249``).  Training-only keys are kept so that user
scripts passing them keep working, but nothing on the predict path reads them.
"""
from __future__ import annotations

from difflib import get_close_matches
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Union

# key -> default (same keys and defaults as the reference default.yaml)
DEFAULT_CFG_DICT: Dict[str, object] = dict(
    task="detect", mode="train",
    model=None, data=None, epochs=100, time=None, patience=100, batch=16, imgsz=640, save=True, save_period=-1,
    device=None, workers=8, project=None, name=None, exist_ok=False, pretrained=True, optimizer="auto", verbose=True,
    seed=0, deterministic=True, single_cls=False, rect=False, cos_lr=False, close_mosaic=10, resume=False, amp=False,
    profile=False, freeze=None, multi_scale=False,
    val=True, split="val", save_json=False, save_hybrid=False, conf=None, iou=0.7, max_det=300, half=False,
    dnn=False, plots=True,
    source=None, vid_stride=1, stream_buffer=False, visualize=False, augment=False, agnostic_nms=False, classes=None,
    embed=None,
    show=False, save_frames=False, save_txt=False, save_conf=False, save_crop=False, show_labels=True,
    show_conf=True, show_boxes=True, line_width=None,
    lr0=0.01, lrf=0.01, momentum=0.937, weight_decay=0.0005, warmup_epochs=3.0, warmup_momentum=0.8,
    warmup_bias_lr=0.1, box=7.5, cls=0.5, dfl=1.5, label_smoothing=0.0, nbs=64, hsv_h=0.015, hsv_s=0.7, hsv_v=0.4,
    degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0, flipud=0.0, fliplr=0.5, bgr=0.0, mosaic=1.0,
    mixup=0.0, copy_paste=0.0, copy_paste_mode="flip", auto_augment="randaugment",
    cfg=None,
)
DEFAULT_CFG_KEYS = DEFAULT_CFG_DICT.keys()

CFG_FLOAT_KEYS = {"warmup_epochs", "box", "cls", "dfl", "degrees", "shear", "time", "workspace", "batch"}
CFG_FRACTION_KEYS = {"dropout", "lr0", "lrf", "momentum", "weight_decay", "warmup_momentum", "warmup_bias_lr",
                     "label_smoothing", "hsv_h", "hsv_s", "hsv_v", "translate", "scale", "perspective", "flipud",
                     "fliplr", "bgr", "mosaic", "mixup", "copy_paste", "conf", "iou", "fraction"}
CFG_INT_KEYS = {"epochs", "patience", "workers", "seed", "close_mosaic", "mask_ratio", "max_det", "vid_stride",
                "line_width", "nbs", "save_period"}
CFG_BOOL_KEYS = {"save", "exist_ok", "verbose", "deterministic", "single_cls", "rect", "cos_lr", "overlap_mask",
                 "val", "save_json", "save_hybrid", "half", "dnn", "plots", "show", "save_txt", "save_conf",
                 "save_crop", "save_frames", "show_labels", "show_conf", "visualize", "augment", "agnostic_nms",
                 "retina_masks", "show_boxes", "keras", "optimize", "int8", "dynamic", "simplify", "nms", "profile",
                 "multi_scale"}


class IterableSimpleNamespace(SimpleNamespace):
    """Namespace that iterates as (key, value) pairs and offers ``get`` (reference utils/__init__.py:292-331)."""

    def __iter__(self):
        return iter(vars(self).items())

    def __str__(self):
        return "\n".join(f"{k}={v}" for k, v in vars(self).items())

    def __getattr__(self, attr):
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def get(self, key, default=None):
        return getattr(self, key, default)


DEFAULT_CFG = IterableSimpleNamespace(**DEFAULT_CFG_DICT)


def cfg2dict(cfg):
    if isinstance(cfg, (str, Path)):
        import yaml

        with open(cfg, errors="ignore", encoding="utf-8") as f:
            cfg = yaml.safe_load(f) or {}
    elif isinstance(cfg, SimpleNamespace):
        cfg = vars(cfg)
    return cfg


def check_dict_alignment(base: Dict, custom: Dict, e=None):
    """Unknown keys raise SyntaxError with close-match suggestions (reference cfg/__init__.py:219-249)."""
    mismatched = [k for k in custom if k not in base]
    if mismatched:
        msg = ""
        for x in mismatched:
            matches = get_close_matches(x, list(base.keys()))
            matches = [f"{k}={base[k]}" if base.get(k) is not None else k for k in matches]
            hint = f"Similar arguments are i.e. {matches}." if matches else ""
            msg += f"'{x}' is not a valid YOLO argument. {hint}\n"
        raise SyntaxError(msg) from e


def get_cfg(cfg: Union[str, Path, Dict, SimpleNamespace] = DEFAULT_CFG_DICT, overrides: Dict = None):
    """Merge ``overrides`` into ``cfg`` and type-check (reference cfg/__init__.py:125-179)."""
    cfg = dict(cfg2dict(cfg))
    if overrides:
        overrides = dict(cfg2dict(overrides))
        if "save_dir" not in cfg:
            overrides.pop("save_dir", None)
        check_dict_alignment(cfg, overrides)
        cfg = {**cfg, **overrides}
    for k in ("project", "name"):
        if k in cfg and isinstance(cfg[k], (int, float)):
            cfg[k] = str(cfg[k])
    if cfg.get("name") == "model":
        cfg["name"] = str(cfg.get("model", "")).split(".")[0]
    for k, v in cfg.items():
        if v is None:
            continue
        if k in CFG_FLOAT_KEYS and not isinstance(v, (int, float)):
            raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}. "
                            f"Valid '{k}' types are int (i.e. '{k}=0') or float (i.e. '{k}=0.5')")
        elif k in CFG_FRACTION_KEYS:
            if not isinstance(v, (int, float)):
                raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}. "
                                f"Valid '{k}' types are int (i.e. '{k}=0') or float (i.e. '{k}=0.5')")
            if not (0.0 <= v <= 1.0):
                raise ValueError(f"'{k}={v}' is an invalid value. Valid '{k}' values are between 0.0 and 1.0.")
        elif k in CFG_INT_KEYS and not isinstance(v, int):
            raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}. '{k}' must be an int (i.e. '{k}=8')")
        elif k in CFG_BOOL_KEYS and not isinstance(v, bool):
            raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}. "
                            f"'{k}' must be a bool (i.e. '{k}=True' or '{k}=False')")
    return IterableSimpleNamespace(**cfg)


def increment_path(path, exist_ok=False, sep="", mkdir=False):
    """runs/detect/predict -> runs/detect/predict2 ... (reference utils/files.py increment_path)."""
    path = Path(path)
    if path.exists() and not exist_ok:
        path, suffix = (path.with_suffix(""), path.suffix) if path.is_file() else (path, "")
        for n in range(2, 9999):
            p = f"{path}{sep}{n}{suffix}"
            if not Path(p).exists():
                break
        path = Path(p)
    if mkdir:
        path.mkdir(parents=True, exist_ok=True)
    return path


def get_save_dir(args, name=None) -> Path:
    """Output directory derived from project/name/task/mode (reference cfg/__init__.py:182-195)."""
    if getattr(args, "save_dir", None):
        return Path(args.save_dir)
    project = args.project or Path("runs") / args.task
    name = name or args.name or f"{args.mode}"
    return increment_path(Path(project) / name, exist_ok=args.exist_ok)
