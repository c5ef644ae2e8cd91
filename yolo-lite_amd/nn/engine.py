"""HipEngine: owns one lowered program + its workspace on a GPU and drives libyl_amd.so.

PyTorch is used only for device memory (the workspace / IO tensors) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from .arch import LayerSpec
from .lower import BF16, F32, FP8, Program, TView, lower, plane_pixels

_TORCH2YL = {torch.float32: _lib.YL_F32, torch.float16: _lib.YL_F16, torch.uint8: _lib.YL_U8}


def _cview(v: Optional[TView]) -> _lib.View:
    if v is None:
        return _lib.View(-1, 0, 0, 0, 0, 0, 0)
    return _lib.View(v.offset, v.H, v.W, v.C, v.pitch, v.coff, v.dtype)


class HipEngine:
    """A YOLO11 network instantiated for a fixed (batch, H, W) on one GPU."""

    def __init__(self, prog: Program, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("HipEngine needs a GPU device (there is no CPU fallback for the HIP path)")
        self.lib = _lib.lib()
        self.device = device
        self.batch, self.in_h, self.in_w = prog.batch, prog.in_h, prog.in_w
        batch, in_h, in_w = prog.batch, prog.in_h, prog.in_w
        self.prog = prog
        self.nc = self.prog.nc
        self.anchors = self.prog.anchors
        ops = self.prog.ops
        arr = (_lib.Op * len(ops))()
        self._keep: list = []
        for i, o in enumerate(ops):
            c = arr[i]
            c.kind = o["kind"]
            c.k = o.get("k", 0)
            c.stride = o.get("stride", 1)
            c.act = o.get("act", 0)
            c.accumulate = o.get("accumulate", 0)
            c.heads, c.key_dim, c.head_dim = o.get("heads", 0), o.get("key_dim", 0), o.get("head_dim", 0)
            c.scale = o.get("scale", 0.0)
            c.grp, c.grp_stride, c.grp_off = o.get("grp", 0), o.get("grp_stride", 0), o.get("grp_off", 0)
            c.cin_weight = o.get("cin_weight", 0)
            c.lane = o.get("lane", 0)
            c.group = int(o.get("group", 0))
            c.dfl_level, c.dfl_keep = int(o.get("dfl_level", 0)), int(o.get("dfl_keep", 0))
            c.inp, c.out, c.res = _cview(o["inp"]), _cview(o["out"]), _cview(o.get("res"))
            c.inp2, c.up = _cview(o.get("inp2")), int(o.get("up", 0))
            c.in_scale, c.out_scale, c.res_scale = (float(o.get(k, 0.0)) for k in ("in_scale", "out_scale", "res_scale"))
            c.out8, c.out8_c0 = _cview(o.get("out8")), int(o.get("out8_c0", 0))
            c.out8_scale, c.out8_only = float(o.get("out8_scale", 0.0)), int(o.get("out8_only", 0))
            if "weight" in o:
                w = np.ascontiguousarray(o["weight"], dtype=np.float32)
                b = np.ascontiguousarray(o["bias"], dtype=np.float32)
                self._keep += [w, b]
                c.weight, c.bias = w.ctypes.data, b.ctypes.data
            if "stem_weight" in o:
                w0 = np.ascontiguousarray(o["stem_weight"], dtype=np.float32)
                b0 = np.ascontiguousarray(o["stem_bias"], dtype=np.float32)
                self._keep += [w0, b0]
                c.stem_weight, c.stem_bias, c.stem_act = w0.ctypes.data, b0.ctypes.data, int(o.get("stem_act", 1))
                c.reserved2 = int(o.get("mid_c", 0))
            if "extra_weight" in o:  # ops of more than three convs: arrays of host pointers
                xw = [np.ascontiguousarray(w, dtype=np.float32) for w in o["extra_weight"]]
                xb = [np.ascontiguousarray(b, dtype=np.float32) for b in o["extra_bias"]]
                pw = (C.c_void_p * len(xw))(*[w.ctypes.data for w in xw])
                pb = (C.c_void_p * len(xb))(*[b.ctypes.data for b in xb])
                self._keep += xw + xb + [pw, pb]
                c.n_extra, c.extra_weight, c.extra_bias = len(xw), C.cast(pw, C.c_void_p), C.cast(pb, C.c_void_p)
            # YL_OP_C3K: workgroups per image (0 = the engine's choice); fused stem / C3k2: bit 8 = keep the output cacheable
            c.reserved3 = int(o.get("split", 0)) | (0x100 if o.get("cacheable") else 0)
            if "post_weight" in o:
                w2 = np.ascontiguousarray(o["post_weight"], dtype=np.float32).reshape(o["post_weight"].shape[0], -1)
                b2 = np.ascontiguousarray(o["post_bias"], dtype=np.float32)
                self._keep += [w2, b2]
                c.post_weight, c.post_bias = w2.ctypes.data, b2.ctypes.data
                c.post_cin, c.post_act = int(w2.shape[1]), int(o.get("post_act", 0))
        lv = (_lib.Level * len(self.prog.levels))()
        for i, L in enumerate(self.prog.levels):
            lv[i] = _lib.Level(L["offset"], L["offset_cls"], L["H"], L["W"], L["stride"], L["nc8"], L.get("offset_xywh", -1))
        self._levels = lv
        iv = _cview(self.prog.input_view)
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.yl_engine_create(arr, len(ops), batch, in_h, in_w, C.byref(iv), lv, len(lv), self.nc,
                                                 self.prog.ws_bytes, C.byref(handle)), "yl_engine_create")
        self._h = handle
        self._slots: List[dict] = []
        self.workspace = self._slot(0)["ws"]
        self._keep.clear()  # weights were packed and uploaded by yl_engine_create

    def _slot(self, i: int) -> dict:
        """Slot i = one workspace (+ NMS scratch and static outputs): batches in flight on different HIP
        streams use different slots of ONE engine (shared packed weights), e.g. bench.py pipelines two."""
        while len(self._slots) <= i:
            with torch.cuda.device(self.device):
                # zero the guard pixels of the planar layout once; no kernel ever writes them.  torch.zeros, not
                # empty + prepare alone: the allocator may hand out the address of a workspace this engine has seen
                # (and cleared) before, which yl_engine_prepare_workspace would then skip
                ws = torch.zeros(self.prog.ws_bytes, dtype=torch.uint8, device=self.device)
                _lib.check(self.lib.yl_engine_prepare_workspace(self._h, ws.data_ptr(),
                                                                torch.cuda.current_stream(self.device).cuda_stream),
                           "yl_engine_prepare_workspace")
                torch.cuda.current_stream(self.device).synchronize()
            self._slots.append(dict(ws=ws, static={}))
        return self._slots[i]

    @classmethod
    def from_model(cls, specs: Sequence[LayerSpec], state_dict: Dict[str, torch.Tensor], batch: int, in_h: int,
                   in_w: int, device: torch.device, strides: Sequence[float] = (8.0, 16.0, 32.0),
                   keep_layers: bool = False, fp8_calib: Optional[torch.Tensor] = None,
                   fp8_headroom: float = 2.0, one_batch: bool = False, dfl_epilogue: bool = True,
                   keep_logits: bool = False) -> "HipEngine":
        """``keep_layers=True`` (debug: ``layer_output`` of every layer) disables the fusions that make
        intermediate tensors disappear; pass False for the production program.

        ``fp8_calib`` (a representative input batch) selects the e4m3 mode of BASELINE.json config 5: the bf16 program
        runs once on it, every dense 3x3 conv's input range is measured (``calibrate_fp8``), and the program is
        lowered again with those convs on the fp8 matrix pipe (``lower.apply_fp8``).

        ``one_batch=True``: the latency profile (``lower(one_batch=True)``) for callers that run one batch at a time.

        ``dfl_epilogue`` (default): the Detect levels' boxes are decoded from the fp32 accumulators of the conv that computes the DFL
        logits (``level_boxes()``); ``keep_logits=True`` ALSO stores the 64 bf16 logits per anchor (``raw_levels()``: parity tests);
        ``dfl_epilogue=False`` is the form of rounds 1-5: logits stored in bf16 and decoded by their readers."""
        prog = lower(specs, state_dict, batch, in_h, in_w, strides, fuse_post=not keep_layers, one_batch=one_batch,
                     dfl_epilogue=dfl_epilogue, keep_logits=keep_logits)
        if fp8_calib is None:
            return cls(prog, device)
        e16 = cls(prog, device)
        scales = e16.calibrate_fp8(fp8_calib, fp8_headroom)
        del e16
        torch.cuda.empty_cache()
        return cls(lower(specs, state_dict, batch, in_h, in_w, strides, fuse_post=not keep_layers, fp8_scales=scales, one_batch=one_batch,
                         dfl_epilogue=dfl_epilogue, keep_logits=keep_logits), device)

    def calibrate_fp8(self, im: torch.Tensor, headroom: float = 2.0) -> Dict[str, float]:
        """{conv name: input scale} for every conv the fp8 pass may take: absmax of the conv's input after a forward of
        this (bf16) program on ``im``, times ``headroom``, mapped onto e4m3's largest finite value 448.  Buffers are
        never recycled, so every tensor is still in the workspace afterwards; a buffer that is updated in place (the
        Bottleneck residual stream) is measured in its final, largest state."""
        from .lower import fp8_candidate

        self.forward(im, decode=False)
        torch.cuda.synchronize(self.device)
        scales: Dict[str, float] = {}
        for o in self.prog.ops:
            if fp8_candidate(o, allow_1x1=True):
                v = o["inp"]
                m = float(self._planes(v)[:, :, : v.H, : v.W].abs().amax().float().item())
                scales[o["name"]] = max(m, 1e-6) * headroom / 448.0
        return scales

    def fp8_conv_scales(self) -> Dict[str, float]:
        """{conv name: input scale} of the convs THIS program runs in e4m3 (empty for a bf16 program)."""
        return {o["name"]: float(o["in_scale"]) for o in self.prog.ops if o["kind"] == 1 and o["inp"].dtype == FP8}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self.lib.yl_engine_destroy(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ forward
    def _check_input(self, im: torch.Tensor):
        if tuple(im.shape) != (self.batch, 3, self.in_h, self.in_w):
            raise ValueError(f"engine built for {(self.batch, 3, self.in_h, self.in_w)}, got {tuple(im.shape)}")
        if im.dtype not in _TORCH2YL:
            raise TypeError(f"unsupported input dtype {im.dtype}")
        if im.device != self.device or not im.is_contiguous():
            raise ValueError("input must be a contiguous tensor on the engine's device")

    def forward(self, im: torch.Tensor, y: Optional[torch.Tensor] = None, graph: bool = False,
                decode: bool = True, slot: int = 0) -> Optional[torch.Tensor]:
        """im (B,3,H,W) fp32/fp16 in [0,1] or uint8 -> y (B, 4+nc, A) fp32 (reference head.py:95-126)."""
        self._check_input(im)
        sl = self._slot(slot)
        if decode and y is None:
            if graph:  # graphs bake pointers: reuse one output buffer (valid until the next graph call)
                if "y" not in sl["static"]:
                    sl["static"]["y"] = torch.empty((self.batch, 4 + self.nc, self.anchors), dtype=torch.float32,
                                                    device=self.device)
                y = sl["static"]["y"]
            else:
                y = torch.empty((self.batch, 4 + self.nc, self.anchors), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        fn = self.lib.yl_engine_forward_graph if graph else self.lib.yl_engine_forward
        _lib.check(fn(self._h, im.data_ptr(), _TORCH2YL[im.dtype], sl["ws"].data_ptr(),
                      y.data_ptr() if decode else None, stream), "yl_engine_forward")
        return y if decode else None

    def detect(self, im: torch.Tensor, conf: float = 0.25, iou: float = 0.7, max_det: int = 300,
               classes=None, agnostic: bool = False, max_nms: int = 30000, max_wh: float = 7680.0,
               scale: Optional[torch.Tensor] = None, graph: bool = True, slot: int = 0, stage: bool = False, rec_key=0,
               count_host: bool = False, scale_static: bool = False):
        """Network + fused Detect decode + NMS; returns padded (out[B, max_det, 6], count[B]).

        Bit-identical to ``forward`` followed by ``utils.ops.nms_padded`` but the (B, 4+nc, A) tensor
        of the reference (head.py:126) is never materialised.  No host synchronisation.  With
        ``graph=True`` the whole path is ONE hipGraph launch and the returned tensors are engine-owned
        buffers that stay valid until the next ``detect`` call with the same ``max_det`` and ``slot``.
        Calls with different ``slot`` values may be in flight on different streams at the same time.
        ``stage=True`` (with ``graph=True``): ``im`` is first copied into an engine-owned input buffer of the slot
        (one device-to-device copy), so callers that hand over a fresh tensor every call -- the predictor API -- still
        replay ONE captured graph instead of re-capturing per input address.
        ``stage="rebind"``: no copy either -- the slot's ONE captured graph gets ``im``'s address written into the kernel nodes
        that read the input (``yl_engine_detect_graph_rebind``); the caller keeps ``im`` alive until the batch has finished and
        has at most one batch per slot pending.
        ``rec_key`` (graph mode): which of the slot's engine-owned output records to write -- a slot may own several
        (``dist.ShardedRunner`` alternates two per slot so that the gather of one step can still read its record while
        the slot's next step runs); each (input address, record) pair is its own captured graph.
        ``count_host`` (graph mode): the per-image counts are written by the last kernel straight into PINNED HOST memory
        (the returned ``cnt`` is that host tensor: valid once the stream has been synchronised, no device-to-host copy) --
        what a caller that waits for every batch wants; the packed record's count words are not written then."""
        self._check_input(im)
        rebind = graph and stage == "rebind"  # (no staging copy: the captured graph's input-reading nodes take im's address)
        if graph and stage and not rebind:
            sk = ("in", im.dtype)
            st0 = self._slot(slot)["static"]
            if sk not in st0:
                st0[sk] = torch.empty_like(im)
            if st0[sk].data_ptr() != im.data_ptr():
                st0[sk].copy_(im, non_blocking=True)
            im = st0[sk]
        max_cand = min(int(max_nms), self.anchors)
        sl = self._slot(slot)
        st = sl["static"]
        wk = ("work", max_cand)
        if wk not in st:  # NMS scratch belongs to the slot: two batches in flight must not share it
            nbytes = int(self.lib.yl_nms_workspace_bytes(self.batch, self.anchors, max_cand))
            st[wk] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        work = st[wk]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if graph:
            ok = (max_det, rec_key)
            if ok not in st:
                # one packed record (boxes, then the counts as int32 bit patterns): the multi-GPU gather sends it as
                # ONE message (dist.pack_views / dist.gather_record)
                from ..dist import pack_views, record_floats

                rec = torch.empty(record_floats(self.batch, max_det), dtype=torch.float32, device=self.device)
                st[("rec", max_det, rec_key)] = rec
                st[ok] = pack_views(rec, self.batch, max_det)
            out, cnt = st[ok]
            if count_host:
                hk = ("cnt_host", max_det, rec_key)
                if hk not in st:  # (hipHostMalloc'ed memory is device-accessible at its host address)
                    st[hk] = torch.zeros(self.batch, dtype=torch.int32, pin_memory=True)
                cnt = st[hk]
        else:
            out = torch.empty((self.batch, max_det, 6), dtype=torch.float32, device=self.device)
            cnt = torch.empty((self.batch,), dtype=torch.int32, device=self.device)
        cls_t = None
        if classes is not None:
            ck = ("cls", tuple(int(c) for c in np.atleast_1d(classes)))
            if ck not in st:  # graphs bake the pointer: one device copy per distinct class filter
                st[ck] = torch.as_tensor(list(ck[1]), dtype=torch.int32, device=self.device)
            cls_t = st[ck]
        if scale is not None and graph:
            if "scale" not in st:
                st["scale"] = torch.empty((self.batch, 5), dtype=torch.float32, device=self.device)
            # (scale_static: the caller promises never to modify this tensor -- the predictor hands over ONE cached table per
            # shape tuple -- so the same object is not copied again: a device-to-device copy in front of every launch otherwise)
            if not (scale_static and st.get("scale_src") is scale):
                st["scale"].copy_(scale.reshape(self.batch, 5))
                st["scale_src"] = scale if scale_static else None
            scale = st["scale"]
        args = (float(conf), float(iou), int(max_det), int(bool(agnostic)),
                cls_t.data_ptr() if cls_t is not None else None, int(cls_t.numel()) if cls_t is not None else 0,
                max_cand, float(max_wh), scale.data_ptr() if scale is not None else None, out.data_ptr(),
                cnt.data_ptr(), work.data_ptr(), stream)
        if graph == "prepare":  # the captured graph's handle, not launched (prepare_detect / launch_many)
            h = C.c_void_p()
            rc = self.lib.yl_engine_detect_graph_prepare(self._h, im.data_ptr(), _TORCH2YL[im.dtype], sl["ws"].data_ptr(),
                                                         *args, C.byref(h))
            _lib.check(rc, "yl_engine_detect_graph_prepare")
            return out, cnt, h.value
        if rebind:
            rc = self.lib.yl_engine_detect_graph_rebind(self._h, im.data_ptr(), _TORCH2YL[im.dtype], sl["ws"].data_ptr(),
                                                        *args)
            _lib.check(rc, "yl_engine_detect_graph_rebind")
        elif graph:
            rc = self.lib.yl_engine_detect_graph(self._h, im.data_ptr(), _TORCH2YL[im.dtype], sl["ws"].data_ptr(),
                                                 *args)
            _lib.check(rc, "yl_engine_detect_graph")
        else:
            self.forward(im, graph=False, decode=False, slot=slot)
            rc = self.lib.yl_decode_nms(sl["ws"].data_ptr(), self._levels, len(self._levels), self.batch, self.nc,
                                        *args)
            _lib.check(rc, "yl_decode_nms")
        return out, cnt

    def prepare_detect(self, im: torch.Tensor, conf: float = 0.25, iou: float = 0.7, max_det: int = 300, slot: int = 0, **kw):
        """``detect(graph=True)`` without the launch: returns ``(out, cnt, graph_handle)`` -- the slot's engine-owned result
        buffers and the handle of the captured hipGraph of exactly this call (same input tensor, thresholds, slot) for
        ``launch_many``.  The first call for a given argument tuple runs the path once on the current stream and captures it."""
        return self.detect(im, conf, iou, max_det=max_det, graph="prepare", slot=slot, **kw)

    def launch_many(self, handles: Sequence[int], streams: Sequence["torch.cuda.Stream"]) -> None:
        """Issue prepared steps back to back from native code (``yl_graph_launch_many``): ``handles[i]`` on ``streams[i]``.
        One call for a whole round of batches in flight instead of one trip through Python per batch (~80 us each, which in
        a 20-step run on 20 streams is a 1.6 ms fill ramp in front of 13 ms of GPU work).  No host synchronisation."""
        n = len(handles)
        if n != len(streams):
            raise ValueError("one stream per step")
        key = (tuple(handles), tuple(s.cuda_stream for s in streams))
        cache = self.__dict__.setdefault("_many", {})
        arr = cache.get(key)
        if arr is None:
            if len(cache) > 16:
                cache.clear()
            arr = cache[key] = ((C.c_void_p * n)(*handles), (C.c_void_p * n)(*[s.cuda_stream for s in streams]))
        _lib.check(self.lib.yl_graph_launch_many(arr[0], arr[1], n), "yl_graph_launch_many")

    def detect_record(self, im: torch.Tensor, conf: float = 0.25, iou: float = 0.7, max_det: int = 300, slot: int = 0,
                      rec_key=0, **kw) -> torch.Tensor:
        """``detect(graph=True)`` returning the slot's packed result record (``dist.record_floats`` fp32 words:
        ``batch * max_det * 6`` box words, then the per-image counts as int32 bit patterns) -- what one rank
        contributes to the per-step gather of the sharded path."""
        self.detect(im, conf, iou, max_det=max_det, graph=True, slot=slot, rec_key=rec_key, **kw)
        return self._slot(slot)["static"][("rec", max_det, rec_key)]

    def level_views(self, i: int):
        """(box, cls) arena views of Detect level i (bf16 planar buffers of 64 and nc8 channels)."""
        L = self.prog.levels[i]
        return (TView(L["offset"], L["H"], L["W"], 64, 64, 0, BF16),
                TView(L["offset_cls"], L["H"], L["W"], L["nc8"], L["nc8"], 0, BF16))

    def raw_levels(self) -> List[torch.Tensor]:
        """(B, 64+nc, H, W) fp32 copies of the Detect level maps (reference head.py:64-65 ``cat(cv2, cv3)``).  The 64 DFL logits
        exist only in a program lowered with ``keep_logits=True`` (the production program decodes the boxes in the epilogue of the
        conv that computes them and stores ``level_boxes()`` instead)."""
        out = []
        for i in range(len(self.prog.levels)):
            if self.prog.levels[i]["offset"] < 0:
                raise RuntimeError("this program does not store the DFL logits (lower(..., keep_logits=True) does); "
                                   "see level_boxes() / raw_cls_levels()")
            vb, vc = self.level_views(i)
            out.append(torch.cat([self.view_tensor(vb), self.view_tensor(vc)[:, : self.nc]], 1))
        return out

    def raw_cls_levels(self) -> List[torch.Tensor]:
        """(B, nc, H, W) fp32 copies of the Detect levels' class logits (cv3, head.py:64)."""
        return [self.view_tensor(self.level_views(i)[1])[:, : self.nc] for i in range(len(self.prog.levels))]

    def head_outputs(self) -> List[torch.Tensor]:
        """Everything the Detect head of this program stores, level by level (class logits; the DFL logits where kept; the decoded
        boxes where the epilogue produces them): what two variants of a program must agree on bit for bit (tests)."""
        out = self.raw_cls_levels()
        for i, L in enumerate(self.prog.levels):
            if L["offset"] >= 0:
                out.append(self.view_tensor(self.level_views(i)[0]))
        if all(L.get("offset_xywh", -1) >= 0 for L in self.prog.levels):
            out += self.level_boxes()
        return out

    def level_boxes(self) -> List[torch.Tensor]:
        """(B, 4, H * W) fp32 per level: (cx, cy, w, h) in input pixels as the epilogue of the level's last box conv decoded them
        from its fp32 accumulators (``yl_level.offset_xywh``; reference head.py:105-106 ``dbox * strides``, per level)."""
        out = []
        for L in self.prog.levels:
            off = L.get("offset_xywh", -1)
            if off < 0:
                raise RuntimeError("this program stores DFL logits, not decoded boxes (lower(..., dfl_epilogue=False))")
            n = self.batch * L["H"] * L["W"]
            t = self.workspace[off: off + 16 * n].view(torch.float32).view(self.batch, L["H"] * L["W"], 4)
            out.append(t.permute(0, 2, 1).contiguous())
        return out

    def _planes(self, v: TView) -> torch.Tensor:
        """The guarded planes of a bf16 view as a (C/8, B, H+1, W+1, 8) tensor aliasing the workspace
        (index [..., :H, :W, :] is the data, the rest are guard pixels); fp8 views: (C/16, B, H+1, W+1, 16) e4m3."""
        pp = plane_pixels(self.batch, v.H, v.W)
        if v.dtype == FP8:
            t = self.workspace[v.offset: v.offset + (v.pitch // 16) * pp * 16].view(torch.float8_e4m3fn).view(v.pitch // 16, pp, 16)
            t = t[v.coff // 16: (v.coff + v.C) // 16, v.W + 2: v.W + 2 + self.batch * (v.H + 1) * (v.W + 1)]
            return t.view(v.C // 16, self.batch, v.H + 1, v.W + 1, 16)
        n = (v.pitch // 8) * pp * 8
        t = self.workspace[v.offset: v.offset + 2 * n].view(torch.bfloat16).view(v.pitch // 8, pp, 8)
        t = t[v.coff // 8: (v.coff + v.C) // 8, v.W + 2: v.W + 2 + self.batch * (v.H + 1) * (v.W + 1)]
        return t.view(v.C // 8, self.batch, v.H + 1, v.W + 1, 8)

    def view_tensor(self, v: TView) -> torch.Tensor:
        """Copy of an arena view as an NCHW fp32 tensor (debug / per-layer parity tests)."""
        if v.dtype == F32:
            n = self.batch * v.H * v.W * v.pitch
            t = self.workspace[v.offset: v.offset + 4 * n].view(torch.float32).view(self.batch, v.H, v.W, v.pitch)
            return t[..., v.coff: v.coff + v.C].permute(0, 3, 1, 2).contiguous()
        t = self._planes(v)[:, :, : v.H, : v.W]  # (C/8, B, H, W, 8); fp8: (C/16, B, H, W, 16), raw e4m3 values (no scale)
        return t.float().permute(1, 0, 4, 2, 3).reshape(self.batch, v.C, v.H, v.W).contiguous()

    def write_view(self, v: TView, x: torch.Tensor) -> None:
        """Store an NCHW tensor into a bf16 arena view (tests / micro-benchmarks); fp8 views take e4m3-representable values."""
        if v.dtype == FP8:
            t = x.to(self.device, torch.float32).view(self.batch, v.C // 16, 16, v.H, v.W).permute(1, 0, 3, 4, 2)
            self._planes(v).view(torch.uint8)[:, :, : v.H, : v.W] = t.contiguous().to(torch.float8_e4m3fn).view(torch.uint8)
            return
        t = x.to(self.device, torch.bfloat16).view(self.batch, v.C // 8, 8, v.H, v.W).permute(1, 0, 3, 4, 2)
        self._planes(v)[:, :, : v.H, : v.W] = t

    def guards_are_zero(self, v: TView) -> bool:
        """True if no kernel has written a guard pixel of this view (layout invariant, tests)."""
        t = self._planes(v)
        if v.dtype == FP8:
            t = t.view(torch.uint8)
        return bool((t[:, :, v.H, :] == 0).all()) and bool((t[:, :, :, v.W] == 0).all())

    def layer_output(self, i: int) -> torch.Tensor:
        return self.view_tensor(self.prog.layer_views[i])

    def op_kernel(self, i: int) -> str:
        """Name of the kernel family the engine chose for op i (yl_engine_op_info)."""
        fl, by, nm = C.c_double(), C.c_double(), C.c_char_p()
        self.lib.yl_engine_op_info(self._h, i, C.byref(fl), C.byref(by), C.byref(nm))
        return nm.value.decode()

    def profile_detect(self, im: torch.Tensor, conf: float = 0.25, iou: float = 0.7, max_det: int = 300,
                       iters: int = 20) -> List[dict]:
        """Per-launch HIP-event timings of the hot path as ``detect`` runs it (yl_engine_profile_detect): every network
        op, then the kernels that really follow it -- class pre-filter, exact decode, sort, mask, scan -- with
        algorithmic flops / bytes per launch."""
        self._check_input(im)
        n = self.lib.yl_engine_num_ops(self._h)
        ms, mp = (C.c_float * n)(), (C.c_float * 5)()
        max_cand = min(30000, self.anchors)  # as detect(): the reference's max_nms cap (ops.py:254-255)
        sl = self._slot(0)
        wk = ("work", max_cand)
        if wk not in sl["static"]:
            sl["static"][wk] = torch.empty(int(self.lib.yl_nms_workspace_bytes(self.batch, self.anchors, max_cand)),
                                           dtype=torch.uint8, device=self.device)
        out = torch.empty((self.batch, max_det, 6), dtype=torch.float32, device=self.device)
        cnt = torch.empty((self.batch,), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.yl_engine_profile_detect(self._h, im.data_ptr(), _TORCH2YL[im.dtype], sl["ws"].data_ptr(),
                                                     float(conf), float(iou), int(max_det), max_cand, out.data_ptr(),
                                                     cnt.data_ptr(), sl["static"][wk].data_ptr(), stream, iters, ms, mp),
                   "yl_engine_profile_detect")
        rows = []
        fl, by, nm = C.c_double(), C.c_double(), C.c_char_p()
        for i in range(n):
            self.lib.yl_engine_op_info(self._h, i, C.byref(fl), C.byref(by), C.byref(nm))
            kern = nm.value.decode()
            if kern == "(in the group launch)" and rows:  # a member of the group launch in front of it: one row per LAUNCH
                rows[-1]["name"] += " | " + self.prog.ops[i].get("name", "")
                rows[-1]["ms"] += ms[i]  # (the event pair around nothing: ~5 us of timing overhead, counted like every row's)
                rows[-1]["flops"] += fl.value
                rows[-1]["bytes"] += by.value
                continue
            nbytes = by.value
            if i == 0 and kern.startswith("stem"):
                # yl_engine_op_info prices the network input at 4 bytes per value (its dtype is a call-time argument): the
                # launch that reads THIS input reads im.element_size() bytes per value
                nbytes -= float(im.numel()) * (4 - im.element_size())
            rows.append(dict(name=self.prog.ops[i].get("name", ""), kernel=kern, ms=ms[i], flops=fl.value, bytes=nbytes))
        ncand = float(cnt.sum().item())  # detections, a lower bound of the candidates the later stages touch
        nc8 = self.prog.levels[0]["nc8"] if self.prog.levels else self.nc
        post = [("detect.prefilter", "decode_candidates_sparse", float(self.batch * self.anchors * nc8 * 2)),
                ("detect.decode_exact", "decode_exact", ncand * (64 + nc8) * 2 + ncand * 32),
                ("nms.sort", "nms_sort", ncand * 16 * 2),
                ("nms.mask", "nms_mask", ncand * 16 + ncand * 8),
                ("nms.scan", "nms_scan", ncand * 16 + float(self.batch * max_det * 24))]
        fused_pf = bool(self.lib.yl_engine_prefilter_fused(self._h, float(conf)))
        env = os.environ.get("YL_NMS_FUSED")  # nms.hip: decode_nms_fused_kernel from 16 images per batch up
        one_launch = (env != "0" if env is not None else self.batch >= 16) and conf >= 0.05
        for (name, kern, b), t in zip(post, mp):
            if fused_pf and name == "detect.prefilter":
                continue  # runs in the epilogue of the launches that write the class logits: not a launch of its own
            if one_launch and name != "detect.prefilter" and rows and rows[-1]["kernel"] == "decode_nms_fused":
                rows[-1]["ms"] += t  # exact decode + sort + mask + scan are ONE launch: one row
                rows[-1]["bytes"] += b
                continue
            if one_launch and name == "detect.decode_exact":
                name, kern = "detect.decode_exact+nms.sort+mask+scan", "decode_nms_fused"
            rows.append(dict(name=name, kernel=kern, ms=t, flops=0.0, bytes=b))
        return rows

    def profile(self, im: torch.Tensor, iters: int = 20) -> List[dict]:
        """Per-op HIP-event timings with algorithmic flops/bytes (see yl_engine_profile)."""
        self._check_input(im)
        n = self.lib.yl_engine_num_ops(self._h)
        ms = (C.c_float * (n + 2))()
        y = torch.empty((self.batch, 4 + self.nc, self.anchors), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.yl_engine_profile(self._h, im.data_ptr(), _TORCH2YL[im.dtype], self.workspace.data_ptr(),
                                              y.data_ptr(), stream, iters, ms), "yl_engine_profile")
        rows = [dict(name="input", kernel="convert_input", ms=ms[0], flops=0.0,
                     bytes=float(im.numel() * im.element_size() + self.batch * self.in_h * self.in_w * 16))]
        fl, by, nm = C.c_double(), C.c_double(), C.c_char_p()
        for i in range(n):
            self.lib.yl_engine_op_info(self._h, i, C.byref(fl), C.byref(by), C.byref(nm))
            rows.append(dict(name=self.prog.ops[i].get("name", ""), kernel=nm.value.decode(), ms=ms[i + 1],
                             flops=fl.value, bytes=by.value))
        rows.append(dict(name="decode", kernel="detect_decode", ms=ms[n + 1], flops=0.0,
                         bytes=float(self.batch * self.anchors * ((64 + self.nc) + (4 + self.nc)) * 4)))
        return rows
