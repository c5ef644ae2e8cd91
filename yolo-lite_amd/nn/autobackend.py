"""AutoBackend: device placement and forward dispatch (reference ``yololite/nn/autobackend.py``).

The backend is always the HIP engine.  Engines are built lazily per (batch, H, W) because all
buffer offsets, tilings and the captured graph are shape-specialised; the most recent few are cached.
"""
from __future__ import annotations

import os

from collections import OrderedDict
from pathlib import Path
from typing import Optional

import torch

from .engine import HipEngine
from .tasks import DetectionModel, attempt_load_weights


def default_class_names(data=None):
    if data:
        try:
            import yaml

            with open(data, errors="ignore", encoding="utf-8") as f:
                return yaml.safe_load(f)["names"]
        except Exception:
            pass
    return {i: f"class{i}" for i in range(999)}


class AutoBackend:
    """``AutoBackend(weights, device, fp16, fuse)`` with ``.forward(im)``, ``.warmup(imgsz)`` and the
    attributes the predictor reads: ``pt fp16 stride names device`` (reference autobackend.py:124)."""

    MAX_ENGINES = 4

    def __init__(self, weights="yolo11n.pt", device=torch.device("cpu"), dnn=False, data=None, fp16=False, batch=1,
                 fuse=True, verbose=True):
        from .. import _lib

        # The one runtime knob left is GPU_MAX_HW_QUEUES, and it only matters to callers that keep SEVERAL batches in
        # flight on several streams (one hardware queue per stream: _lib.RUNTIME_KNOBS).  predict() runs one stream, so
        # AutoBackend does not set it; a caller that pipelines exports it (or calls _lib.apply_runtime_knobs()) before
        # its first GPU call.  All that is done here is to say so when such a request came too late to take effect.
        st = _lib.runtime_knob_status()
        if verbose and st["applied"] and st["before_hip_init"] is False:
            from ..utils import LOGGER

            LOGGER.warning("GPU_MAX_HW_QUEUES was set after the HIP runtime had initialised (%s): streams beyond the runtime's "
                           "default of four hardware queues will share queues; set it before the first GPU call", st)
        w = weights[0] if isinstance(weights, (list, tuple)) else weights
        if isinstance(w, DetectionModel):
            model = w
        else:
            # reference bug not inherited: `w.split('.')[1] == 'pt'` (autobackend.py:63) breaks on './x.pt'
            if Path(str(w)).suffix not in (".pt", ".safetensors"):
                raise TypeError(f"model='{w}' is not a supported model format (only *.pt / *.safetensors / in-memory models)")
            model = attempt_load_weights(str(w), device=device, inplace=True, fuse=fuse)
        if fuse:
            model = model.fuse(verbose=verbose)
        self.model = model
        self.device = device if isinstance(device, torch.device) else torch.device(device)
        self.fp16 = bool(fp16)  # accepted for API compatibility; the engine computes in bf16 / fp32 accumulate
        self.pt = True
        self.nn_module = isinstance(w, DetectionModel)
        self.jit = False
        self.triton = False
        self.stride = max(int(model.stride.max()), 32)
        self.names = model.names if model.names else default_class_names(data)
        self.task = "detect"
        self.batch = batch
        self._engines: "OrderedDict[tuple, HipEngine]" = OrderedDict()

    def eval(self):
        return self

    def engine_for(self, b: int, h: int, w: int, calib: "torch.Tensor | None" = None, one_batch: bool = False) -> HipEngine:
        """The engine for this input shape (built on first use).  ``YL_DTYPE=fp8`` selects the e4m3 mode (dense 3x3 convs
        on the fp8 matrix pipe); its activation scales are calibrated on ``calib``, the first batch
        of that shape the predictor sees."""
        # (one_batch: the predictor's per-call path runs ONE batch at a time and takes the latency profile of the program,
        # its pipelined path the throughput profile; same kernels, same bits)
        key = (b, h, w, self.model._version, bool(one_batch))
        eng = self._engines.get(key)
        if eng is None:
            # one profile of a shape alive at a time: with both resident, predict(stream=True) fell from 52 k to 46 k images/s (an
            # idle engine's streams, graphs and 2-42 GB of workspaces are not free on this runtime); a caller that alternates
            # modes on one shape pays a rebuild (~0.5 s) per switch
            if os.environ.get("YL_ENGINE_ONE_PROFILE", "1") == "1":
                if self._engines.pop((b, h, w, self.model._version, not one_batch), None) is not None:
                    torch.cuda.empty_cache()
            if self.device.type != "cuda":
                raise RuntimeError("the YOLO11 network runs on the GPU only (HIP kernels; no CPU fallback): "
                                   "pass device=0 / 'cuda:0'")
            fp8 = os.environ.get("YL_DTYPE", "bf16").lower() == "fp8"
            if fp8 and calib is None:
                raise RuntimeError("YL_DTYPE=fp8 needs a calibration batch: call forward() / predict() with real inputs")
            eng = HipEngine.from_model(self.model.specs, self.model.state_dict(), b, h, w, self.device,
                                       strides=self.model.stride.tolist(), fp8_calib=calib if fp8 else None, one_batch=one_batch)
            self._engines[key] = eng
            while len(self._engines) > self.MAX_ENGINES:
                self._engines.popitem(last=False)
        else:
            self._engines.move_to_end(key)
        return eng

    def forward(self, im: torch.Tensor, augment=False, visualize=False, embed=None, graph: bool = False):
        """(B,3,H,W) in [0,1] -> [y (B, 4+nc, A) fp32, [per-level class-logit maps (B, nc, H_l, W_l)]]
        like ``Detect.forward`` in eval mode (reference head.py:59-69, autobackend.py:126-151; see the note at the return)."""
        if augment or visualize or embed:
            raise NotImplementedError("augment / visualize / embed are outside the accelerated predict path")
        if im.device != self.device:
            im = im.to(self.device)
        if im.dtype == torch.float64 or im.dtype == torch.bfloat16:
            im = im.float()
        im = im.contiguous()
        b, _, h, w = im.shape
        eng = self.engine_for(b, h, w, calib=im)
        y = eng.forward(im, graph=graph)
        # The second element: the reference returns the raw level maps (B, 64 + nc, H_l, W_l) (head.py:69).  This program decodes
        # the boxes in the epilogue of the conv that computes the 64 DFL logits and never stores them: the levels' class logits
        # (B, nc, H_l, W_l) are returned in their place (nothing on the predict path reads it, predictor.py:99 takes preds[0]).
        return [y, eng.raw_cls_levels()]

    __call__ = forward

    def profile_layers(self, im: torch.Tensor, iters: int = 10, log: bool = True):
        """Per-LAYER profile of one batch, the table ``BaseModel._profile_one_layer`` prints with ``predict(profile=True)`` (reference
        nn/tasks.py:155-177: ``time (ms)  GFLOPs  params  module`` per layer of the yaml, then the total): HIP-event time of every
        launch of the hot path (``yl_engine_profile_detect``), summed per yolo11.yaml layer.  A launch that fuses several layers (the
        stem launch = layers 0 + 1 + the next C3k2's cv1) is charged to its first layer; ``Detect`` gets its branches' launches and
        ``(post)`` the decode + NMS launch, which the reference times outside the model.  Returns the rows as dicts."""
        import re

        from ..utils import LOGGER
        from .arch import conv_units

        if im.device != self.device:
            im = im.to(self.device)
        im = im.contiguous()
        b, _, h, w = im.shape
        eng = self.engine_for(b, h, w, calib=im)
        specs = self.model.specs
        params = {sp.i: 0 for sp in specs}
        for u in conv_units(specs):
            li = int(u.name.split(".")[1])
            params[li] += u.c2 * (u.c1 // u.g) * u.k * u.k + (2 * u.c2 if u.bn else u.c2)
        per = {sp.i: dict(layer=sp.i, module=sp.kind, ms=0.0, gflops=0.0, params=params[sp.i], launches=0) for sp in specs}
        post = dict(layer="post", module="decode + NMS", ms=0.0, gflops=0.0, params=0, launches=0)
        for r in eng.profile_detect(im, iters=iters):
            m = re.match(r"model\.(\d+)", r["name"])
            row = per[int(m.group(1))] if m else post
            row["ms"] += r["ms"]
            row["gflops"] += r["flops"] / 1e9
            row["launches"] += 1
        rows = [per[sp.i] for sp in specs] + [post]
        if log:
            LOGGER.info(f"{'time (ms)':>10s} {'GFLOPs':>10s} {'params':>10s}  module")
            for r in rows:
                LOGGER.info(f"{r['ms']:10.3f} {r['gflops']:10.2f} {r['params']:10.0f}  {r['layer']}: {r['module']} ({r['launches']} launches)")
            LOGGER.info(f"{sum(r['ms'] for r in rows):10.3f} {'-':>10s} {'-':>10s}  Total (batch {b}, {h}x{w}, in-sequence HIP events)")
        return rows

    def from_numpy(self, x):
        import numpy as np

        return torch.tensor(x).to(self.device) if isinstance(x, np.ndarray) else x

    def warmup(self, imgsz=(1, 3, 640, 640)):
        """One dummy forward on GPU devices (reference autobackend.py:157-165, minus its `self.jit` bug)."""
        if os.environ.get("YL_DTYPE", "bf16").lower() == "fp8":
            return  # the e4m3 engine is calibrated on the first real batch, not on a dummy one
        if self.device.type != "cpu":
            im = torch.zeros(*imgsz, dtype=torch.float32, device=self.device)
            self.forward(im, graph=False)
