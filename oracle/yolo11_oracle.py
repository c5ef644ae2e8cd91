"""CPU oracle for the YOLO11 detect inference path  --  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 PyTorch-CPU functional ops and numpy, what the
reference (dongyunjinyu/YOLO-Lite @ 2025-03-03) computes on the path
``YOLOLite.predict -> DetectionModel._predict_once -> Detect._inference ->
ops.non_max_suppression -> ops.scale_boxes``.  It is the checker the HIP path is
compared against.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing in ``yolo-lite_amd/``
does, and the product path raises if the HIP library is missing instead of
falling back to this code.

Pinning status
--------------
* network forward / Detect decode / scale_boxes / LetterBox geometry: PINNED against
  golden vectors produced by importing the reference itself in the build container
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked in
  ``tests/test_oracle_golden.py``).
* ``torchvision.ops.nms`` (call site reference ``yololite/utils/ops.py:265``) is a
  third-party dependency that is absent from /root/reference and from this image, and
  the reference pins no version (no requirements/lock file).  ``nms_greedy`` below
  restates its published semantics (sort by score descending, keep greedily, suppress
  j when IoU(i, j) > thr, strict).  PARITY UNPINNED at that boundary: no reference
  test or vector exists for it; ties between exactly equal scores are broken here by
  lower candidate index first (stable sort).
* ``cv2.resize`` pixels in LetterBox are likewise third-party and unpinned; only the
  letterbox geometry (sizes, pads, gain) is restated and pinned.

Every function cites the reference file:line it follows (paths relative to
/root/reference/yololite).

bf16-storage mode (``forward(..., storage="bf16")``)
----------------------------------------------------
The SAME restatement with every tensor that the HIP path keeps in HBM / LDS as bf16 rounded to bf16 at the point
where the HIP path rounds it (the network input, BN-folded filters, every conv / depthwise / attention output after
bias + activation + residual, the softmax probabilities entering the P.V product, the Detect levels' class logits -- and, with
``box_logits="bf16"`` only, their 64 DFL logits: the product of rounds 1-5 stored those as bf16, since round 6 it decodes the boxes
from the fp32 accumulators of the conv that computes them, which ``box_logits="fp32"``, the default, mirrors); all
arithmetic stays fp32.  It splits the HIP path's distance from the fp32 reference into (a) the error of the number
format, a CPU-only property measured in ``tests/test_oracle_golden.py`` against the reference-pinned fp32 mode, and
(b) the kernels' own deviation (accumulation order, hardware exp / rcp), which the GPU tests bound tightly against
this mode.  fp32 mode is the pinned one; bf16 mode differs from it ONLY by those roundings (and by folding the BN
into the filters first, utils/torch_utils.py:182-209, as the product does).

e4m3 emulation (``forward(..., storage="bf16", fp8_scales={conv name: input scale})``)
--------------------------------------------------------------------------------------
BASELINE.json config 5 runs the dense 3x3 convs in OCP e4m3 (csrc/conv_f8.hip).  For the convs named in ``fp8_scales`` the
bf16-storage mode additionally rounds what that kernel rounds: the conv's INPUT to e4m3(x / scale) (saturating at 448, round to
nearest even) with the calibrated per-tensor scale, and its BN-folded FILTER to e4m3 per output channel (scale = absmax / 448,
conv_f8_pack_weights); the products and sums stay fp32, as on the matrix pipe.  It measures what the e4m3 FORMAT costs
against the pinned fp32 mode, with no kernel involved -- the yardstick for the HIP e4m3 program (tests/test_gpu_fp8.py,
bench.py ``format_error_e4m3``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # utils/torch_utils.py:249 (initialize_weights sets every BatchNorm2d eps)

# cfg/yolo11.yaml:7-12  [depth, width, max_channels]
SCALES = {
    "n": (0.50, 0.25, 1024),
    "s": (0.50, 0.50, 1024),
    "m": (0.50, 1.00, 512),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.50, 512),
}

# cfg/yolo11.yaml:15-47 as (from, repeats, kind, args)
_BACKBONE_HEAD = [
    (-1, 1, "Conv", (64, 3, 2)),
    (-1, 1, "Conv", (128, 3, 2)),
    (-1, 2, "C3k2", (256, False, 0.25)),
    (-1, 1, "Conv", (256, 3, 2)),
    (-1, 2, "C3k2", (512, False, 0.25)),
    (-1, 1, "Conv", (512, 3, 2)),
    (-1, 2, "C3k2", (512, True)),
    (-1, 1, "Conv", (1024, 3, 2)),
    (-1, 2, "C3k2", (1024, True)),
    (-1, 1, "SPPF", (1024, 5)),
    (-1, 2, "C2PSA", (1024,)),
    (-1, 1, "Upsample", ()),
    ((-1, 6), 1, "Concat", ()),
    (-1, 2, "C3k2", (512, False)),
    (-1, 1, "Upsample", ()),
    ((-1, 4), 1, "Concat", ()),
    (-1, 2, "C3k2", (256, False)),
    (-1, 1, "Conv", (256, 3, 2)),
    ((-1, 13), 1, "Concat", ()),
    (-1, 2, "C3k2", (512, False)),
    (-1, 1, "Conv", (512, 3, 2)),
    ((-1, 10), 1, "Concat", ()),
    (-1, 2, "C3k2", (1024, True)),
    ((16, 19, 22), 1, "Detect", ()),
]


def make_divisible(x: float, divisor: int) -> int:
    """utils/ops.py:101-114."""
    return math.ceil(x / divisor) * divisor


def plan(scale: str, nc: int = 80, ch: int = 3) -> List[dict]:
    """Channel / repeat arithmetic of nn/tasks.py:525-664 (parse_model) for yolo11."""
    depth, width, max_ch = SCALES[scale]
    chs: List[int] = []
    out: List[dict] = []
    c_prev = ch
    for i, (f, n, kind, args) in enumerate(_BACKBONE_HEAD):
        n = max(round(n * depth), 1) if n > 1 else n  # tasks.py:558
        c1 = c_prev if f == -1 else None
        if kind in ("Conv", "C3k2", "SPPF", "C2PSA"):
            c1 = chs[f] if (f != -1 and i > 0) else c_prev
            c2 = make_divisible(min(args[0], max_ch) * width, 8)  # tasks.py:596
            spec = dict(i=i, f=f, kind=kind, c1=c1, c2=c2, n=n)
            if kind == "Conv":
                spec.update(k=args[1], s=args[2])
            elif kind == "C3k2":
                c3k = bool(args[1]) or scale in "mlx"  # tasks.py:622-625
                e = args[2] if len(args) > 2 else 0.5
                spec.update(c3k=c3k, e=e)
        elif kind == "Upsample":
            c2 = c_prev
            spec = dict(i=i, f=f, kind=kind, c1=c_prev, c2=c2)
        elif kind == "Concat":
            c2 = sum(c_prev if j == -1 else chs[j] for j in f)
            spec = dict(i=i, f=f, kind=kind, c2=c2)
        elif kind == "Detect":
            spec = dict(i=i, f=f, kind=kind, ch=[chs[j] for j in f], nc=nc, c2=None)
            c2 = None
        else:  # pragma: no cover
            raise ValueError(kind)
        out.append(spec)
        chs.append(c2)
        c_prev = c2
    return out


# --------------------------------------------------------------------------- modules
_STORAGE = ["fp32"]  # "fp32" (the pinned restatement) or "bf16" (storage emulation, see the module docstring)
_FP8_SCALES: Dict[str, float] = {}  # bf16 mode only: convs emulated in e4m3 (name -> calibrated input scale)
_BOX_LOGITS = ["fp32"]  # bf16 mode only: "bf16" also rounds the Detect levels' DFL logits (the stored-logits product of rounds 1-5)
E4M3_MAX = 448.0


def e4m3_round(t: torch.Tensor) -> torch.Tensor:
    """Saturating round-to-nearest-even onto the OCP e4m3 grid (v_cvt_pk_fp8_f32 behind a clamp, csrc/conv_common.h pack_f8x4)."""
    return t.float().clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float()


def q_store(t: torch.Tensor) -> torch.Tensor:
    """A tensor as the HIP path stores it: rounded to bf16 (round to nearest even) in bf16-storage mode."""
    return t.bfloat16().float() if _STORAGE[0] == "bf16" else t


def conv_bn_act(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, k: int = 1, s: int = 1,
                g: int = 1, act: bool = True, res: Optional[torch.Tensor] = None, dw_taps_bf16: bool = False) -> torch.Tensor:
    """nn/modules/conv.py:35-49 Conv.forward: SiLU(BN(conv2d(x))), bias-free conv,
    pad = k//2 (autopad conv.py:26-32), BN in eval mode with eps 1e-3.  ``res``: a shortcut added to the result
    (block.py:343, 946-947); in fp32 mode ``conv(x) + res`` is exactly the reference's ``x + conv(x)``."""
    if _STORAGE[0] == "bf16":
        w, b = fuse_conv_bn(sd[p + ".conv.weight"], sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                            sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"])
        # dense filters are packed as bf16; depthwise taps stay fp32 (csrc/engine.hip d_w) except in the Detect class
        # branch, whose depthwise convs run on the matrix pipe with bf16 taps (csrc/conv_mfma.hip dw_pack_mfma)
        if g == 1 or dw_taps_bf16:
            w = w.bfloat16().float()
        if p in _FP8_SCALES and g == 1:  # an e4m3 conv of config 5: input and filter as csrc/conv_f8.hip sees them
            sc = float(_FP8_SCALES[p])
            x = e4m3_round(x / sc) * sc
            m = w.abs().flatten(1).amax(1)
            ws = torch.where(m > 0, m / E4M3_MAX, torch.ones_like(m)).view(-1, 1, 1, 1)
            w = e4m3_round(w / ws) * ws
        y = F.conv2d(x, w, b, s, k // 2, 1, g)
        y = F.silu(y) if act else y
        return q_store(y if res is None else y + res)
    y = F.conv2d(x, sd[p + ".conv.weight"], None, s, k // 2, 1, g)
    y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"],
                     sd[p + ".bn.weight"], sd[p + ".bn.bias"], False, 0.0, BN_EPS)
    y = F.silu(y) if act else y
    return y if res is None else res + y


def plain_conv(sd, p, x, store: bool = True):
    """nn.Conv2d with bias, no BN / activation (the last conv of each Detect branch, head.py:38-48).  ``store=False``: the result
    is consumed from the fp32 accumulators (bf16-storage mode does not round it)."""
    w = sd[p + ".weight"]
    y = F.conv2d(x, w.bfloat16().float() if _STORAGE[0] == "bf16" else w, sd[p + ".bias"])
    return q_store(y) if store else y


def bottleneck(sd, p, x, c1, c2, shortcut=True, k=(3, 3), e=0.5):
    """nn/modules/block.py:330-343."""
    y = conv_bn_act(sd, p + ".cv1", x, k[0], 1)
    return conv_bn_act(sd, p + ".cv2", y, k[1], 1, res=x if (shortcut and c1 == c2) else None)


def c3k(sd, p, x, c, n=2):
    """nn/modules/block.py:245-259 (C3.forward) with block.py:731-739 (C3k: Bottleneck k=(3,3), e=1.0)."""
    a = conv_bn_act(sd, p + ".cv1", x, 1, 1)
    for j in range(n):
        a = bottleneck(sd, f"{p}.m.{j}", a, c // 2, c // 2, True, (3, 3), 1.0)
    b = conv_bn_act(sd, p + ".cv2", x, 1, 1)
    return conv_bn_act(sd, p + ".cv3", torch.cat((a, b), 1), 1, 1)


def c3k2(sd, p, x, c2, n, use_c3k, e):
    """nn/modules/block.py:220-235 (C2f.forward) + block.py:720-728 (C3k2)."""
    c = int(c2 * e)
    y = list(conv_bn_act(sd, p + ".cv1", x, 1, 1).chunk(2, 1))
    for j in range(n):
        if use_c3k:
            y.append(c3k(sd, f"{p}.m.{j}", y[-1], c, 2))
        else:
            y.append(bottleneck(sd, f"{p}.m.{j}", y[-1], c, c, True, (3, 3), 0.5))
    return conv_bn_act(sd, p + ".cv2", torch.cat(y, 1), 1, 1)


def sppf(sd, p, x):
    """nn/modules/block.py:165-184: three chained MaxPool2d(5,1,2)."""
    y = [conv_bn_act(sd, p + ".cv1", x, 1, 1)]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], 5, 1, 2))
    return conv_bn_act(sd, p + ".cv2", torch.cat(y, 1), 1, 1)


def attention(sd, p, x, num_heads, residual=False):
    """nn/modules/block.py:863-916 Attention.forward (attn_ratio 0.5); ``residual``: return x + attn(x) (block.py:946)."""
    B, C, H, W = x.shape
    N = H * W
    hd = C // num_heads
    kd = int(hd * 0.5)
    scale = kd ** -0.5
    qkv = conv_bn_act(sd, p + ".qkv", x, 1, 1, 1, act=False)
    q, k, v = qkv.view(B, num_heads, kd * 2 + hd, N).split([kd, kd, hd], dim=2)
    attn = (q.transpose(-2, -1) @ k) * scale
    if _STORAGE[0] == "bf16":
        # csrc/ops_misc.hip attention_kernel: exp(s - max) enters the P.V product as bf16, the row sum stays fp32
        e = (attn - attn.amax(-1, keepdim=True)).exp()
        o = q_store((v @ q_store(e).transpose(-2, -1)) / e.sum(-1).unsqueeze(-2)).view(B, C, H, W)
        # the depthwise `pe` conv accumulates onto the stored attention output (yl_op.accumulate)
        o = conv_bn_act(sd, p + ".pe", v.reshape(B, C, H, W), 3, 1, C, act=False, res=o)
        return conv_bn_act(sd, p + ".proj", o, 1, 1, 1, act=False, res=x if residual else None)
    attn = attn.softmax(dim=-1)
    o = (v @ attn.transpose(-2, -1)).view(B, C, H, W)
    o = o + conv_bn_act(sd, p + ".pe", v.reshape(B, C, H, W), 3, 1, C, act=False)
    y = conv_bn_act(sd, p + ".proj", o, 1, 1, 1, act=False)
    return x + y if residual else y


def psablock(sd, p, x, c):
    """nn/modules/block.py:919-953: x = x + attn(x); x = x + ffn(x) (the adds run in the producing conv's epilogue
    in the product, hence through ``res=`` here)."""
    x = attention(sd, p + ".attn", x, c // 64, residual=True)
    f = conv_bn_act(sd, p + ".ffn.0", x, 1, 1)
    return conv_bn_act(sd, p + ".ffn.1", f, 1, 1, 1, act=False, res=x)


def c2psa(sd, p, x, c1, n):
    """nn/modules/block.py:999-1038."""
    c = int(c1 * 0.5)
    a, b = conv_bn_act(sd, p + ".cv1", x, 1, 1).split((c, c), dim=1)
    for j in range(n):
        b = psablock(sd, f"{p}.m.{j}", b, c)
    return conv_bn_act(sd, p + ".cv2", torch.cat((a, b), 1), 1, 1)


def make_anchors(shapes: Sequence[Tuple[int, int]], strides: Sequence[float], offset: float = 0.5):
    """utils/tal.py:326-338: anchors (x+0.5, y+0.5) row-major per level."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, dtype=torch.float32) + offset
        sy = torch.arange(h, dtype=torch.float32) + offset
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((gx, gy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=torch.float32))
    return torch.cat(pts), torch.cat(st)


def dfl(box: torch.Tensor, reg_max: int = 16) -> torch.Tensor:
    """nn/modules/block.py:51-69: softmax over 16 bins, expectation with arange(16)."""
    b, _, a = box.shape
    p = box.view(b, 4, reg_max, a).transpose(2, 1).softmax(1)
    w = torch.arange(reg_max, dtype=torch.float32).view(1, reg_max, 1, 1)
    return (p * w).sum(1)


def dist2bbox_xywh(dist: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
    """utils/tal.py:341-350 with xywh=True, dim=1."""
    lt, rb = dist.chunk(2, 1)
    x1y1 = anchors - lt
    x2y2 = anchors + rb
    return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1)


def detect_heads(sd, p, feats: List[torch.Tensor], nc: int = 80) -> List[torch.Tensor]:
    """nn/modules/head.py:29-69: per level cat(cv2[i](x), cv3[i](x)) -> (B, 64+nc, H, W)."""
    outs = []
    for i, x in enumerate(feats):
        b = conv_bn_act(sd, f"{p}.cv2.{i}.0", x, 3, 1)
        b = conv_bn_act(sd, f"{p}.cv2.{i}.1", b, 3, 1)
        b = plain_conv(sd, f"{p}.cv2.{i}.2", b, store=_BOX_LOGITS[0] == "bf16")
        c = conv_bn_act(sd, f"{p}.cv3.{i}.0.0", x, 3, 1, x.shape[1], dw_taps_bf16=True)
        c = conv_bn_act(sd, f"{p}.cv3.{i}.0.1", c, 1, 1)
        c = conv_bn_act(sd, f"{p}.cv3.{i}.1.0", c, 3, 1, c.shape[1], dw_taps_bf16=True)
        c = conv_bn_act(sd, f"{p}.cv3.{i}.1.1", c, 1, 1)
        c = plain_conv(sd, f"{p}.cv3.{i}.2", c)
        outs.append(torch.cat((b, c), 1))
    return outs


def detect_decode(raw: List[torch.Tensor], strides: Sequence[float], nc: int = 80) -> torch.Tensor:
    """nn/modules/head.py:95-126 Detect._inference (non-export branch)."""
    B = raw[0].shape[0]
    no = nc + 64
    x_cat = torch.cat([r.reshape(B, no, -1) for r in raw], 2)
    anchors, st = make_anchors([r.shape[2:] for r in raw], strides)
    anchors, st = anchors.transpose(0, 1), st.transpose(0, 1)
    box, cls = x_cat.split((64, nc), 1)
    dbox = dist2bbox_xywh(dfl(box), anchors.unsqueeze(0)) * st
    return torch.cat((dbox, cls.sigmoid()), 1)


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, scale: str = "n", nc: int = 80,
            strides: Sequence[float] = (8.0, 16.0, 32.0), return_layers: bool = False, storage: str = "fp32",
            fp8_scales: Optional[Dict[str, float]] = None, box_logits: str = "fp32"):
    """nn/tasks.py:118-145 BaseModel._predict_once over the 24 layers of yolo11.yaml.

    Returns (y[B, 4+nc, A], [raw level maps (B, 64+nc, H, W)]) like Detect.forward (head.py:59-69).
    ``storage="bf16"``: the bf16-storage emulation described in the module docstring; with ``fp8_scales`` the named convs are
    emulated in e4m3 on top of it; ``box_logits="bf16"`` additionally rounds the levels' DFL logits (see the module docstring).
    """
    if storage not in ("fp32", "bf16"):
        raise ValueError(storage)
    if fp8_scales and storage != "bf16":
        raise ValueError("the e4m3 emulation sits on top of the bf16-storage mode")
    if box_logits not in ("fp32", "bf16"):
        raise ValueError(box_logits)
    prev, prev8, prevb = _STORAGE[0], dict(_FP8_SCALES), _BOX_LOGITS[0]
    _STORAGE[0] = storage
    _BOX_LOGITS[0] = box_logits
    _FP8_SCALES.clear()
    _FP8_SCALES.update(fp8_scales or {})
    try:
        return _forward(sd, q_store(x), scale, nc, strides, return_layers)
    finally:
        _STORAGE[0] = prev
        _BOX_LOGITS[0] = prevb
        _FP8_SCALES.clear()
        _FP8_SCALES.update(prev8)


def _forward(sd, x, scale, nc, strides, return_layers):
    specs = plan(scale, nc)
    ys: List[Optional[torch.Tensor]] = []
    for sp in specs:
        i, f, kind = sp["i"], sp["f"], sp["kind"]
        p = f"model.{i}"
        if f != -1:
            x = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
        if kind == "Conv":
            x = conv_bn_act(sd, p, x, sp["k"], sp["s"])
        elif kind == "C3k2":
            x = c3k2(sd, p, x, sp["c2"], sp["n"], sp["c3k"], sp["e"])
        elif kind == "SPPF":
            x = sppf(sd, p, x)
        elif kind == "C2PSA":
            x = c2psa(sd, p, x, sp["c1"], sp["n"])
        elif kind == "Upsample":
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        elif kind == "Concat":
            x = torch.cat(x, 1)
        elif kind == "Detect":
            raw = detect_heads(sd, p, list(x), nc)
            y = detect_decode(raw, strides, nc)
            if return_layers:
                return y, raw, ys
            return y, raw
        ys.append(x)
    raise RuntimeError("no Detect layer")


# ------------------------------------------------------------------------ post-process
def xywh2xyxy(x: torch.Tensor) -> torch.Tensor:
    """utils/ops.py:372-389."""
    y = torch.empty_like(x)
    xy = x[..., :2]
    wh = x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def nms_greedy(boxes: np.ndarray, scores: np.ndarray, iou_thres: float) -> np.ndarray:
    """Published semantics of torchvision.ops.nms (third party, absent; PARITY UNPINNED).

    fp32 throughout: area = (x2-x1)*(y2-y1); inter = max(0, min(x2)-max(x1)) * max(0, ...);
    iou = inter / (a_i + a_j - inter); j is suppressed when iou > thr (strict).  Returns kept
    indices in descending-score order; equal scores keep their input order (stable).
    """
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes[:, k] for k in range(4))
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    thr = np.float32(iou_thres)
    dead = np.zeros(len(order), dtype=bool)
    keep = []
    for a in range(len(order)):
        if dead[a]:
            continue
        i = order[a]
        keep.append(i)
        rest = order[a + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), (xx2 - xx1).astype(np.float32))
        h = np.maximum(np.float32(0), (yy2 - yy1).astype(np.float32))
        inter = (w * h).astype(np.float32)
        union = ((areas[i] + areas[rest]).astype(np.float32) - inter).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = (inter / union).astype(np.float32)
        dead[a + 1:] |= iou > thr
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(prediction: torch.Tensor, conf_thres: float = 0.25, iou_thres: float = 0.45,
                        classes: Optional[Sequence[int]] = None, agnostic: bool = False,
                        max_det: int = 300, max_nms: int = 30000, max_wh: int = 7680,
                        multi_label: bool = False) -> List[torch.Tensor]:
    """utils/ops.py:138-273 (no masks, not rotated, no `labels`, no wall-clock limit — the reference's
    time-limit break, ops.py:269-271, is a load-dependent defect we do not restate).  ``multi_label`` is the
    validator's mode (engine/validator.py:281-291): every (anchor, class) pair above the threshold is a row
    (ops.py:239-241), in ``torch.where`` order, and ties in the cap / NMS sorts keep that order (stable).

    prediction: (B, 4+nc, A) fp32, rows [cx, cy, w, h, scores...].  Not modified (the reference
    converts in place, ops.py:212-213).
    """
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1  # ops.py:186-187
    prediction = prediction.detach().float().cpu().clone()
    bs = prediction.shape[0]
    nc = prediction.shape[1] - 4
    xc = prediction[:, 4:4 + nc].amax(1) > conf_thres  # ops.py:203
    prediction = prediction.transpose(-1, -2).contiguous()
    prediction[..., :4] = xywh2xyxy(prediction[..., :4])  # ops.py:212-213
    output = [torch.zeros((0, 6))] * bs
    for xi in range(bs):
        x = prediction[xi][xc[xi]]  # ops.py:222
        if not x.shape[0]:
            continue
        box, cls = x[:, :4], x[:, 4:]
        if multi_label and nc > 1:  # ops.py:208, 239-241
            i, j = torch.where(cls > conf_thres)
            x = torch.cat((box[i], x[i, 4 + j, None], j[:, None].float()), 1)
        else:
            conf, j = cls.max(1, keepdim=True)  # first max wins on ties, ops.py:243
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]  # ops.py:244
        if classes is not None:
            x = x[(x[:, 5:6] == torch.tensor(classes, dtype=torch.float32)).any(1)]  # ops.py:247-248
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:
            x = x[x[:, 4].argsort(descending=True, stable=True)[:max_nms]]  # ops.py:254-255
        c = x[:, 5:6] * (0 if agnostic else max_wh)  # ops.py:258
        boxes = (x[:, :4] + c).numpy()  # ops.py:264 (fp32 add: boxes are quantised at large offsets)
        keep = nms_greedy(boxes, x[:, 4].numpy(), iou_thres)[:max_det]  # ops.py:265-266
        output[xi] = x[torch.from_numpy(keep)]
    return output


def clip_boxes(boxes: torch.Tensor, shape: Tuple[int, int]) -> torch.Tensor:
    """utils/ops.py:276-295."""
    boxes[..., 0] = boxes[..., 0].clamp(0, shape[1])
    boxes[..., 1] = boxes[..., 1].clamp(0, shape[0])
    boxes[..., 2] = boxes[..., 2].clamp(0, shape[1])
    boxes[..., 3] = boxes[..., 3].clamp(0, shape[0])
    return boxes


def scale_boxes_params(img1_shape, img0_shape) -> Tuple[float, Tuple[int, int]]:
    """gain / pad arithmetic of utils/ops.py:80-88 (Python banker's round)."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1),
           round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    return gain, pad


def scale_boxes(img1_shape, boxes: torch.Tensor, img0_shape) -> torch.Tensor:
    """utils/ops.py:66-98 (padding=True, xywh=False); modifies and returns ``boxes``."""
    gain, pad = scale_boxes_params(img1_shape, img0_shape)
    boxes[..., 0] -= pad[0]
    boxes[..., 1] -= pad[1]
    boxes[..., 2] -= pad[0]
    boxes[..., 3] -= pad[1]
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


def letterbox_geometry(shape: Tuple[int, int], new_shape=(640, 640), auto: bool = False,
                       stride: int = 32, center: bool = True, scaleup: bool = True) -> dict:
    """Geometry of data/augment.py:636-681 LetterBox.__call__ (no pixels: cv2 is third party).

    Returns r, new_unpad (w, h), pads (top, bottom, left, right) and the output (h, w).
    """
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride  # np.mod, augment.py:654-655
    if center:
        dw /= 2
        dh /= 2
    top, bottom = (int(round(dh - 0.1)) if center else 0), int(round(dh + 0.1))
    left, right = (int(round(dw - 0.1)) if center else 0), int(round(dw + 0.1))
    return dict(r=r, new_unpad=new_unpad, pads=(top, bottom, left, right),
                out_shape=(new_unpad[1] + top + bottom, new_unpad[0] + left + right))


def fuse_conv_bn(w, gamma, beta, mean, var, eps: float = BN_EPS):
    """utils/torch_utils.py:182-209 folding formula (dead code in the reference; the build folds at load)."""
    s = gamma / torch.sqrt(var + eps)
    return w * s.view(-1, 1, 1, 1), beta - mean * s
