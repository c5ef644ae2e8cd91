"""yolo-lite_amd: MI355X-native YOLO11 detect inference behind the YOLO-Lite / Ultralytics-style API.

Drop-in for ``yololite`` on the predict path (reference ``yololite/__init__.py``, ``engine/model.py``):

    from yololite_amd import YOLOLite          # or: YOLO
    model = YOLOLite("yolo11n.pt")             # or "yolo11n.yaml"
    results = model(["boats.jpg"])             # list[Results]

Compute runs in hand-written HIP kernels for gfx950 (``csrc/``) reached through the C ABI in
``include/yl_amd.h``; there is no CPU fallback.
"""
__version__ = "0.1.0"

_LAZY = {
    "YOLOLite": ("engine.model", "YOLOLite"),
    "YOLO": ("engine.model", "YOLO"),
    "DetectionPredictor": ("engine.predictor", "DetectionPredictor"),
    "Results": ("engine.results", "Results"),
    "Boxes": ("engine.results", "Boxes"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = tuple(_LAZY) + ("__version__",)
