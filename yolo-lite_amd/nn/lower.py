"""Lowering of the YOLO11 layer list to a flat engine program over one workspace arena.

Reference behaviour being lowered (all under /root/reference/yololite):
``nn/tasks.py:118-145`` (layer loop and skip list), ``nn/modules/block.py`` (C3k2/C3k/Bottleneck/SPPF/
C2PSA/PSABlock/Attention forward methods), ``nn/modules/head.py:59-69`` (Detect.forward) and
``nn/modules/conv.py:321-331`` (Concat).  ``chunk``/``split``/``cat`` never move data here: producers
write straight into channel slices of the consumer's buffer (activations are stored as planes of 8
channels -- see ``plane_pixels`` -- so a slice is a range of planes),
Conv+BN are folded with the formula of ``utils/torch_utils.py:182-209`` (eps 1e-3, ``:249``) and the
Bottleneck / PSABlock residual adds run in the producing conv's epilogue.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .arch import LayerSpec, detect_mid_channels

BN_EPS = 1e-3
BF16, F32, FP8 = 0, 1, 4  # yl_amd.h element types (FP8 = OCP e4m3, planes of 16 channels)
OP_CONV, OP_DWCONV3, OP_SPPF_POOL, OP_UPSAMPLE, OP_ATTENTION, OP_STEM, OP_C3K2S, OP_BNECK2, OP_QUANT, OP_C3K, OP_PSA, OP_SPPF = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12
OP_PSA_TAIL = 13


@dataclass
class TView:
    offset: int
    H: int
    W: int
    C: int
    pitch: int
    coff: int = 0
    dtype: int = BF16

    def slice(self, coff: int, C: int) -> "TView":
        assert 0 <= coff and coff + C <= self.C, (coff, C, self.C)
        return TView(self.offset, self.H, self.W, C, self.pitch, self.coff + coff, self.dtype)


@dataclass
class Program:
    batch: int
    in_h: int
    in_w: int
    nc: int
    ops: List[dict] = field(default_factory=list)
    ws_bytes: int = 0
    input_view: Optional[TView] = None
    levels: List[dict] = field(default_factory=list)
    layer_views: Dict[int, TView] = field(default_factory=dict)
    anchors: int = 0
    strides: Tuple[float, ...] = ()


def plane_pixels(batch: int, H: int, W: int) -> int:
    """Pixels per plane of the C8-planar guarded bf16 layout (mirror of ``geo_of`` in csrc/yl_internal.h).

    buffer[C/8][plane_pixels][8]: one plane per 8 channels; inside a plane the whole batch is one array
    of rows, each row followed by a zero guard pixel and each image by a zero guard row, so that every
    3x3 tap of every pixel is a constant offset that lands on data or on a zero:
    ``pixel(b, y, x) = g0 + b * (H + 1) * (W + 1) + y * (W + 1) + x`` with ``g0 = W + 2``.
    """
    n = (W + 2) + batch * (H + 1) * (W + 1) + 1
    return (n + 15) // 16 * 16


def view_bytes(batch: int, v: "TView") -> int:
    """Bytes of the whole buffer underneath a view."""
    if v.dtype == F32:
        return batch * v.H * v.W * v.pitch * 4
    return (v.pitch // (16 if v.dtype == FP8 else 8)) * plane_pixels(batch, v.H, v.W) * 16


class _Arena:
    """Bump allocator.  Buffers are never recycled: the guard pixels of the planar layout are zeroed
    once per workspace and must stay untouched (capacity is not a concern with 288 GB of HBM)."""

    def __init__(self, batch: int):
        self.batch = batch
        self.top = 0

    def alloc(self, H: int, W: int, C: int, dtype: int = BF16) -> TView:
        if dtype == FP8 and C % 16:
            raise NotImplementedError(f"fp8 activations need channel counts that are multiples of 16, got {C}")
        if dtype != F32 and C % 8:
            raise NotImplementedError(f"bf16 activations need channel counts that are multiples of 8, got {C}")
        v = TView(self.top, H, W, C, C, 0, dtype)
        self.top += (view_bytes(self.batch, v) + 255) // 256 * 256
        return v

    def alloc_bytes(self, n: int) -> int:
        """A raw buffer of n bytes (no planar geometry): its offset."""
        off = self.top
        self.top += (n + 255) // 256 * 256
        return off


def c3k_resident_fits(c_: int, H: int, W: int) -> bool:
    """Mirror of ``c3k_resident_fits`` (csrc/resident.hip): a C3k block of inner width 64 whose map fits one workgroup's
    LDS -- x as 16 dense planes (later the 8 guarded planes between a Bottleneck's convs) + the 8 guarded planes of `a`."""
    if c_ != 64:
        return False
    ntiles = (H * W + 15) // 16
    xpl = ntiles * 16 * 16
    apl = (((H + 2) * (W + 1) + 2) + 15) // 16 * 16 * 16
    return max(16 * xpl, 8 * apl) + 8 * apl <= 160 * 1024 and 1 <= ntiles <= 28


def psa_resident_fits(c: int, nh: int, kd: int, hd: int, H: int, W: int) -> bool:
    """Mirror of ``psa_resident_fits`` (csrc/resident.hip): the PSABlock of YOLO11n's C2PSA on a map of at most 400 pixels."""
    return c == 128 and nh == 2 and kd == 32 and hd == 64 and 8 <= H * W <= 400


def psa_tail_fits(c: int, c2: int, n_blocks: int) -> bool:
    """Mirror of ``psa_tail_fits`` (csrc/psa_tail.hip): the pointwise tail of a C2PSA layer with one PSABlock of width 128 and
    256 output channels (YOLO11n's layer 10); any map size -- the pixels of the batch are one list."""
    return c == 128 and c2 == 256 and n_blocks == 1


def sppf_resident_fits(c1: int, c2: int, tail: int, H: int, W: int) -> bool:
    """Mirror of ``sppf_resident_fits`` (csrc/sppf_resident.hip): an SPPF layer 256 -> 256 (hidden width 128), optionally with
    the 1x1 conv 256 -> 256 behind it, on a map whose 16 planes + the three-slot filter ring fit one workgroup's LDS."""
    ntiles = (H * W + 15) // 16
    return c1 == 256 and c2 == 256 and tail in (0, 256) and 1 <= ntiles <= 28 and ntiles * 256 * 16 + 3 * 16384 <= 160 * 1024  # (small maps: + 32 KB, < 144 KB)


def fuse_follow_up_convs(ops: List[dict]) -> List[dict]:
    """Conv A -> 1x1 stride-1 conv B, where B is the ONLY reader of A's output and A produces 32 or 64
    channels: B is applied in A's epilogue (``yl_op.post_*``), the tensor between them disappears.
    In YOLO11n: layer 1 -> C3k2.cv1 of layer 2, layer 3 -> C3k2.cv1 of layer 4, and the last two convs
    of every Detect box branch (cv2[i][1] -> cv2[i][2]).  Saves 0.23 GB of HBM traffic per batch of 32
    at 640x640 and five launches."""

    def key(v: TView):
        return (v.offset, v.coff, v.C)

    def planes(v: TView):
        return {(v.offset, p) for p in range(v.coff // 8, (v.coff + v.C) // 8)} if v.dtype == BF16 else set()

    readers: Dict[tuple, List[int]] = {}
    for i, o in enumerate(ops):
        for k in ("inp", "res"):
            v = o.get(k)
            if v is not None and v.dtype == BF16:
                for pl in planes(v):
                    readers.setdefault(pl, []).append(i)
        if o.get("accumulate"):  # reads its own output
            for pl in planes(o["out"]):
                readers.setdefault(pl, []).append(i)
    drop = set()
    for i, a in enumerate(ops):
        if a["kind"] != OP_CONV or i in drop or "post_weight" in a or a.get("res") is not None:
            continue
        y = a["out"]
        if y.dtype != BF16 or y.C not in (32, 64) or a.get("cin_weight", 0):
            continue
        rd = {j for pl in planes(y) for j in readers.get(pl, [])}
        if len(rd) != 1:
            continue
        j = next(iter(rd))
        b = ops[j]
        if (j <= i or b["kind"] != OP_CONV or b.get("k") != 1 or b.get("stride", 1) != 1 or b.get("res") is not None
                or key(b["inp"]) != key(y) or b.get("lane", 0) != a.get("lane", 0) or b["out"].C > y.C
                or "post_weight" in b or b.get("cin_weight", 0)):
            continue
        # nothing between them may write A's output again (in-place updates): A is the only writer by construction
        a["post_weight"] = b["weight"].reshape(b["weight"].shape[0], -1)
        a["post_bias"] = b["bias"]
        a["post_act"] = b.get("act", 0)
        a["out"] = b["out"]
        if b.get("dfl_level", 0):  # (the fused 1x1 computes a Detect level's DFL logits: the box decode moves with it)
            a["dfl_level"], a["dfl_keep"] = b["dfl_level"], b.get("dfl_keep", 0)
        a["name"] = f"{a.get('name', '')}+{b.get('name', '').split('.')[-1] if '.' in b.get('name', '') else b.get('name', '')}"
        a["fused_names"] = [b.get("name", "")]
        drop.add(j)
    return [o for i, o in enumerate(ops) if i not in drop]


def fuse_depthwise_pointwise(ops: List[dict]) -> List[dict]:
    """DWConv 3x3 -> 1x1 Conv (the Detect class branch, head.py:41-48) when the 1x1 conv is the only reader
    of the depthwise output and has at most 80 output channels: one launch, the depthwise output lives in
    LDS only (``yl_op.post_*`` on a YL_OP_DWCONV3 op)."""

    def planes(v: TView):
        return {(v.offset, p) for p in range(v.coff // 8, (v.coff + v.C) // 8)}

    drop = set()
    for i, a in enumerate(ops):
        if a["kind"] != OP_DWCONV3 or a.get("accumulate") or a.get("grp") or "post_weight" in a:
            continue
        if a["inp"].C % 16:  # the fused kernel runs the depthwise conv on 16-channel MFMA tiles
            continue
        mine = planes(a["out"])
        rd = []
        for j, o in enumerate(ops):
            for k in ("inp", "res", "inp2"):
                v = o.get(k)
                if v is not None and v.dtype == BF16 and planes(v) & mine:
                    rd.append(j)
            if o.get("accumulate") and planes(o["out"]) & mine:
                rd.append(j)
        if len(rd) != 1 or rd[0] <= i:
            continue
        b = ops[rd[0]]
        y = a["out"]
        if (b["kind"] != OP_CONV or b.get("k") != 1 or b.get("stride", 1) != 1 or b.get("res") is not None
                or "post_weight" in b or "inp2" in b or b.get("cin_weight", 0)
                or ((b["out"].C + 15) // 16 not in (1, 2, 4, 5)
                    # 128 pointwise channels (YOLO11s' class branch) as eight tiles per wave: built, bit-exact against torch
                    # (tests/test_gpu_kernels.py), and SLOWER than the two launches (116 against 57 + 43 us @80x80 batch 64: a
                    # 64 KB workgroup, one 32 KB slab restaged per 128 pixels): opt-in
                    and not (os.environ.get("YL_FUSE_DWPW128", "0") == "1" and b["out"].C == 128 and a["inp"].C % 32 == 0
                             and a["inp"].C <= 256))
                or b.get("lane", 0) != a.get("lane", 0)
                or (b["inp"].offset, b["inp"].coff, b["inp"].C) != (y.offset, y.coff, y.C)):
            continue
        a["post_weight"] = b["weight"].reshape(b["weight"].shape[0], -1)
        a["post_bias"] = b["bias"]
        a["post_act"] = b.get("act", 0)
        a["out"] = b["out"]
        a["name"] = f"{a.get('name', '')}+pw"
        a["fused_names"] = [b.get("name", "")]
        drop.add(rd[0])
    return [o for i, o in enumerate(ops) if i not in drop]


def fuse_dwpw_tail(ops: List[dict]) -> List[dict]:
    """(DWConv 3x3 -> 1x1 Conv) -> 1x1 Conv, where the last conv is the only reader of the pair's output and has the
    same number of 16-channel tiles (Detect's class branch, head.py:41-48: ..., DWConv(c3, c3, 3), Conv(c3, c3, 1),
    nn.Conv2d(c3, nc, 1)): the last conv joins the fused depthwise + pointwise launch (``yl_op.stem_*`` on a
    YL_OP_DWCONV3 op); the c3-channel tensor between them is never stored."""

    def planes(v: TView):
        return {(v.offset, p) for p in range(v.coff // 8, (v.coff + v.C) // 8)}

    drop = set()
    for i, a in enumerate(ops):
        if a["kind"] != OP_DWCONV3 or "post_weight" not in a or "stem_weight" in a:
            continue
        y = a["out"]
        mine = planes(y)
        rd = [j for j, o in enumerate(ops) for k in ("inp", "res", "inp2")
              if o.get(k) is not None and o[k].dtype == BF16 and planes(o[k]) & mine]
        if len(rd) != 1 or rd[0] <= i:
            continue
        b = ops[rd[0]]
        if (b["kind"] != OP_CONV or b.get("k") != 1 or b.get("stride", 1) != 1 or b.get("res") is not None
                or "post_weight" in b or b.get("inp2") is not None or b.get("cin_weight", 0) or y.C % 8
                or (b["out"].C + 15) // 16 != (y.C + 15) // 16 or b["out"].dtype != BF16
                or b.get("lane", 0) != a.get("lane", 0)
                or (b["inp"].offset, b["inp"].coff, b["inp"].C) != (y.offset, y.coff, y.C)):
            continue
        a["stem_weight"] = b["weight"].reshape(b["weight"].shape[0], -1)
        a["stem_bias"] = b["bias"]
        a["stem_act"] = b.get("act", 0)
        a["mid_c"] = y.C
        a["out"] = b["out"]
        a["name"] = f"{a.get('name', '')}+{b.get('name', '').split('.')[-1]}"
        a["fused_names"] = list(a.get("fused_names", [])) + [b.get("name", "")]
        drop.add(rd[0])
    return [o for i, o in enumerate(ops) if i not in drop]


def fuse_upsample_into_convs(ops: List[dict]) -> List[dict]:
    """nn.Upsample(2, nearest) -> Concat -> 1x1 conv (yolo11.yaml:31-36): when that conv is the only
    reader of the upsampled planes, it reads the low-resolution map itself at pixel (y/2, x/2)
    (``yl_op.in2``); the 4x larger copy is never written or read and the upsample launch disappears."""

    def planes(v: TView):
        return {(v.offset, p) for p in range(v.coff // 8, (v.coff + v.C) // 8)}

    drop = set()
    for i, u in enumerate(ops):
        if u["kind"] != OP_UPSAMPLE or u.get("stride") != 2 or u.get("lane", 0):
            continue
        s_out = u["out"]
        up_planes = planes(s_out)
        rd = []
        for j, o in enumerate(ops):
            for k in ("inp", "res", "inp2"):
                v = o.get(k)
                if v is not None and v.dtype == BF16 and planes(v) & up_planes:
                    rd.append(j)
            if o.get("accumulate") and planes(o["out"]) & up_planes:
                rd.append(j)
        if len(rd) != 1:
            continue
        r = ops[rd[0]]
        v = r["inp"]
        if (r["kind"] != OP_CONV or r.get("k") != 1 or r.get("stride", 1) != 1 or "post_weight" in r or "inp2" in r
                or r.get("cin_weight", 0) or r.get("lane", 0) or v.offset != s_out.offset or v.coff != s_out.coff
                or v.C <= s_out.C or s_out.C % 32 or (v.C - s_out.C) % 32):
            continue
        r["inp2"] = u["inp"]
        r["up"] = 2
        r["inp"] = TView(v.offset, v.H, v.W, v.C - s_out.C, v.pitch, v.coff + s_out.C, v.dtype)
        r["fused_upsample"] = u.get("name", "")
        drop.add(i)
    return [o for i, o in enumerate(ops) if i not in drop]


def fuse_stem_into_conv(ops: List[dict], in_w: int) -> List[dict]:
    """Layer 0 (the stem, op 0) -> layer 1, a 3x3 stride-2 conv that is the stem output's ONLY reader: both run in
    one launch (``yl_op.stem_*``, csrc/stem2.hip) and the stem's output -- the largest tensor of the network, 105 MB
    at batch 32 x 640^2 -- exists only as LDS tiles.  Bit-identical to the two-launch form."""
    if not ops or ops[0]["kind"] != OP_STEM or in_w % 4:
        return ops
    st = ops[0]
    y = st["out"]

    def touches(v):
        return v is not None and v.dtype == BF16 and v.offset == y.offset

    readers = [i for i, o in enumerate(ops) if i and (touches(o.get("inp")) or touches(o.get("res")) or touches(o.get("inp2")))]
    if readers != [1]:
        return ops
    b = ops[1]
    cout1 = b["post_weight"].shape[1] if "post_weight" in b else b["out"].C
    if (b["kind"] != OP_CONV or b.get("k") != 3 or b.get("stride") != 2 or b.get("res") is not None
            or b.get("inp2") is not None or b.get("cin_weight", 0) or b.get("lane", 0)
            or (b["inp"].offset, b["inp"].coff, b["inp"].C) != (y.offset, 0, y.C) or y.pitch != y.C
            or y.C not in ((16, 32) if os.environ.get("YL_FUSE_STEM_WIDE", "1") == "1" else (16,)) or cout1 not in (32, 64)):
        # the 32-channel stem of YOLO11s (bit-identical): neutral in round 2 (206 VGPRs and 75 KB of LDS per workgroup ate
        # what the 0.8 GB less traffic bought), +0.6-1.4 % with the round-3 kernels around it (24.12 -> 24.26-24.48 k images/s
        # at batch 64, one batch at a time +2 %): on; YL_FUSE_STEM_WIDE=0 restores the two launches
        return ops
    b["stem_weight"], b["stem_bias"], b["stem_act"] = st["weight"], st["bias"], st.get("act", 1)
    b["name"] = f"{st.get('name', '')}+{b.get('name', '')}"
    b["fused_names"] = list(b.get("fused_names", [])) + [st.get("name", "")]
    return ops[1:]


def fuse_c3k2_small(ops: List[dict]) -> List[dict]:
    """Bottleneck.cv1 (3x3, c -> c/2), Bottleneck.cv2 (3x3, c/2 -> c, + residual) and C3k2.cv2 (1x1, 3c -> c2) of a
    C3k2 block with c = 16 or 32 (layers 2, 4 and 16 of YOLO11n): one launch (``OP_C3K2S``, csrc/c3k2s.hip) with the
    intermediates in LDS instead of three (0.39 GB per batch of 32 for layer 2 alone).  Bit-identical."""

    def same(u: TView, v: TView):
        return u is not None and v is not None and (u.offset, u.coff, u.C, u.pitch) == (v.offset, v.coff, v.C, v.pitch)

    def reads(o, v: TView):
        for k in ("inp", "res", "inp2"):
            u = o.get(k)
            if u is not None and u.dtype == BF16 and u.offset == v.offset and u.coff < v.coff + v.C and v.coff < u.coff + u.C:
                return True
        return False

    out, i = [], 0
    while i < len(ops):
        a = ops[i]
        b = ops[i + 1] if i + 1 < len(ops) else None
        c = ops[i + 2] if i + 2 < len(ops) else None
        ok = (b is not None and c is not None and all(o["kind"] == OP_CONV and not o.get("lane", 0) and o.get("act", 0)
                                                       and "post_weight" not in o and "stem_weight" not in o
                                                       and o.get("inp2") is None and not o.get("cin_weight", 0) for o in (a, b, c))
              and a.get("k") == 3 and a.get("stride", 1) == 1 and a.get("res") is None and a["inp"].C in (16, 32)
              and a["out"].C * 2 == a["inp"].C
              and b.get("k") == 3 and b.get("stride", 1) == 1 and same(b["inp"], a["out"]) and same(b.get("res"), a["inp"])
              and b["out"].C == a["inp"].C and c.get("k") == 1 and c.get("stride", 1) == 1 and c.get("res") is None
              and c["inp"].C == 3 * a["inp"].C and c["inp"].coff == 0 and c["out"].dtype == BF16
              and (c["out"].C == 64 or (c["out"].C == 128 and a["inp"].C == 32))
              and a["inp"].offset == c["inp"].offset and a["inp"].coff == a["inp"].C and a["inp"].pitch == c["inp"].pitch
              and b["out"].offset == c["inp"].offset and b["out"].coff == 2 * a["inp"].C)
        if ok:  # the intermediate and c_out must have no other reader
            others = [o for j, o in enumerate(ops) if j not in (i, i + 1, i + 2)]
            ok = not any(reads(o, a["out"]) or reads(o, b["out"]) for o in others)
        if ok:
            out.append(dict(kind=OP_C3K2S, name=f"{a.get('name', '')}+cv2+{c.get('name', '').split('.')[-1]}", act=1,
                            inp=c["inp"], out=c["out"], res=None, weight=a["weight"], bias=a["bias"],
                            post_weight=b["weight"], post_bias=b["bias"], stem_weight=c["weight"], stem_bias=c["bias"],
                            fused_names=[b.get("name", ""), c.get("name", "")]))
            i += 3
        else:
            out.append(a)
            i += 1
    return out


def fp8_candidate(o: dict, min_cin: int = 32, allow_1x1: bool = False) -> bool:
    """A conv the fp8 pass may take: a plain dense 3x3 (stride 1 or 2) on bf16 planes whose channel counts fit the
    e4m3 kernel's tiles (csrc/conv_f8.hip).  A 1x1 conv (``allow_1x1``, at least 64 input channels) only pays when its
    e4m3 input costs no pass of its own, i.e. when the tensor has an fp8 mirror (``apply_fp8_mirrors``)."""
    x, y = o.get("inp"), o.get("out")
    if o["kind"] != OP_CONV or not (o.get("k") == 3 or (allow_1x1 and o.get("k") == 1 and x.C >= 64)):
        return False
    return ("post_weight" not in o and o.get("inp2") is None and not o.get("dfl_level", 0)
            and "stem_weight" not in o and not o.get("cin_weight", 0) and x.dtype == BF16 and y.dtype == BF16
            and x.C % 16 == 0 and x.C >= min_cin and (y.C % 32 == 0 or y.C % 48 == 0))


def apply_fp8_mirrors(ops: List[dict], ar: "_Arena", scales: Dict[str, float]) -> List[dict]:
    """e4m3 as a second storage format of the tensors the convs read.  A bf16 buffer whose fp8-capable readers (3x3 AND
    1x1 convs) can all be served gets an fp8 MIRROR: a buffer of the same channel structure (``C16-planar``) and one
    scale, kept in step by every op that writes the bf16 buffer -- each writer's epilogue stores its channel range
    into the mirror too (``yl_op.out8``), or ONLY there when nothing reads the bf16 version any more (an fp8 conv whose
    readers are all fp8 convs).  Concat buffers work because every producer mirrors its own slice; in-place updates
    (the Bottleneck residual stream) work because the updating conv mirrors its result.  Buffers with a writer that
    cannot mirror (a fused block, pooling, attention, upsampling, an odd tile count) are left to ``apply_fp8``."""

    def planes(v: Optional[TView]):
        if v is None or v.dtype != BF16:
            return set()
        return {(v.offset, p) for p in range(v.coff // 8, (v.coff + v.C) // 8)}

    def reads(o: dict):
        r = set()
        for k in ("inp", "res", "inp2"):
            r |= planes(o.get(k))
        if o.get("accumulate") or o["kind"] in (OP_SPPF_POOL,):
            r |= planes(o["out"])
        return r

    # YL_FP8_1X1_GF: 1x1 convs join only above this many GFLOP per launch (default 0: all of them).  Measured on YOLO11x at
    # 1280x1280 batch 8 with 100 (the six C3k2.cv2 convs): 964 images/s against 923, but those convs write the layer outputs and
    # the post-NMS gate already fails (1355 / 2347 boxes matched, max |dconf| 0.145) -- there is no cheap subset.
    min_gf = float(os.environ.get("YL_FP8_1X1_GF", "0"))

    def cand(o: dict) -> bool:
        if not (fp8_candidate(o, allow_1x1=True) and o.get("name") in scales and o["inp"].coff % 16 == 0):
            return False
        if o.get("k") == 1:
            y = o["out"]
            return 2.0 * ar.batch * y.H * y.W * y.C * o["inp"].C >= min_gf * 1e9
        return True

    def can_mirror(o: dict) -> bool:
        y = o["out"]
        if y.coff % 16 or y.C % 16 or "out8" in o:
            return False
        if o["kind"] == OP_STEM:
            return True
        if o["kind"] != OP_CONV or "post_weight" in o or "stem_weight" in o:
            return False
        c16 = (y.C + 15) // 16
        return cand(o) or c16 % 2 == 0

    # buffers (by arena offset) read by candidates, the planes they read, their scale
    want: Dict[int, set] = {}
    scale: Dict[int, float] = {}
    shape: Dict[int, TView] = {}
    for o in ops:
        if cand(o):
            x = o["inp"]
            want.setdefault(x.offset, set()).update(planes(x))
            scale[x.offset] = max(scale.get(x.offset, 0.0), float(scales[o["name"]]))
            shape[x.offset] = x
    mirror: Dict[int, TView] = {}
    for off, pl in want.items():
        wr = [o for o in ops if o["out"].dtype == BF16 and o["out"].offset == off and planes(o["out"]) & pl]
        if wr and all(can_mirror(o) for o in wr) and all((o["out"].H, o["out"].W) == (shape[off].H, shape[off].W) for o in wr):
            x = shape[off]
            mirror[off] = ar.alloc(x.H, x.W, x.pitch, FP8)

    def mview(v: TView) -> TView:
        m = mirror[v.offset]
        return TView(m.offset, v.H, v.W, v.C, m.pitch, v.coff, FP8)

    out: List[dict] = []
    for i, o in enumerate(ops):
        o = dict(o)
        is8 = cand(o) and o["inp"].offset in mirror
        y = o["out"]
        wrote = planes(y)
        if y.dtype == BF16 and y.offset in mirror and wrote & want[y.offset]:
            # does anything still read the bf16 version of what this op writes, before it is overwritten?
            bf16_reader = False
            for r in ops[i + 1:]:
                via_mirror = cand(r) and r["inp"].offset in mirror
                if (planes(r.get("res")) | planes(r.get("inp2"))) & wrote or (planes(r.get("inp")) & wrote and not via_mirror) or \
                        ((r.get("accumulate") or r["kind"] == OP_SPPF_POOL) and planes(r["out"]) & wrote):
                    bf16_reader = True
                    break
                if planes(r.get("out")) >= wrote:
                    break
            if is8 and not bf16_reader:
                o["out"], o["out_scale"] = mview(y), scale[y.offset]
            else:
                o.update(out8=mview(y), out8_c0=0, out8_scale=scale[y.offset])
                if o["kind"] == OP_STEM and not bf16_reader:
                    o["out8_only"] = 1
        if is8:
            o["in_scale"] = scale[o["inp"].offset]
            o["inp"] = mview(o["inp"])
        out.append(o)
    return out


def apply_fp8(ops: List[dict], ar: "_Arena", scales: Dict[str, float]) -> List[dict]:
    """BASELINE.json config 5 ("e4m3 weights + activations, fp32 accumulate"): every candidate conv named in ``scales``
    (its calibrated input scale = absmax x headroom / 448, see ``HipEngine.calibrate_fp8``) runs on the fp8 matrix
    pipe.  Its input is the e4m3 copy of the bf16 tensor -- made once per tensor version by a YL_OP_QUANT op and shared
    by every fp8 reader -- or comes straight from the producing fp8 conv when all readers of that tensor are fp8 convs
    (the first 3x3 of a Bottleneck -> the second): then the bf16 tensor is never written.  Outputs that anything else
    reads, and residuals, stay bf16, so the rest of the program is untouched."""

    def planes(v: Optional[TView]):
        if v is None or v.dtype != BF16:
            return set()
        return {(v.offset, p) for p in range(v.coff // 8, (v.coff + v.C) // 8)}

    def key(v: TView):
        return (v.offset, v.coff, v.C, v.H, v.W)

    def reads(o: dict):
        r = set()
        for k in ("inp", "res", "inp2"):
            r |= planes(o.get(k))
        if o.get("accumulate") or o["kind"] in (OP_SPPF_POOL,):
            r |= planes(o["out"])
        return r

    def cand(o: dict) -> bool:
        return fp8_candidate(o) and o.get("name") in scales

    out: List[dict] = []
    live: Dict[tuple, Tuple[TView, float]] = {}
    dual = os.environ.get("YL_FP8_DUAL", "1") != "0"

    def producer(x: TView, lane: int) -> Optional[dict]:
        """The already emitted op that wrote the current contents of x, if its epilogue can also store the e4m3 copy
        (yl_op.out8): the stem, a plain bf16 conv with an even number of channel tiles per wave, or an fp8 conv."""
        px = planes(x)
        for p in reversed(out):
            po = planes(p.get("out"))
            if not (po & px):
                continue
            if not px <= po or "out8" in p or p.get("lane", 0) != lane or (x.H, x.W) != (p["out"].H, p["out"].W):
                return None
            if p["kind"] == OP_STEM:
                return p if key(x) == key(p["out"]) else None
            if p["kind"] != OP_CONV or "post_weight" in p or "stem_weight" in p:
                return None
            c16 = (p["out"].C + 15) // 16
            if p["inp"].dtype != FP8 and c16 % 2:  # the bf16 kernel writes the copy by pairs of 16-channel tiles
                return None
            return p
        return None

    for i, o in enumerate(ops):
        wrote = planes(o.get("out"))
        if cand(o):
            x = o["inp"]
            if key(x) not in live:
                q = ar.alloc(x.H, x.W, x.C, FP8)
                sc = float(scales[o["name"]])
                p = producer(x, o.get("lane", 0)) if dual else None
                if p is not None:  # the producer's epilogue writes the e4m3 copy next to (or instead of) its bf16 output
                    p.update(out8=q, out8_c0=x.coff - p["out"].coff, out8_scale=sc)
                    # nothing but fp8 convs ever reads this buffer (conservatively: no op of the whole program, before or after):
                    # the copy is the only output -- the stem's (model.0 -> model.1) and, round 6, a plain conv's (YOLO11x: model.2.cv2
                    # -> model.3, 629 MB of bf16 per batch of eight 1280x1280 images that nobody read)
                    only_ok = p["kind"] == OP_STEM or (key(x) == key(p["out"]) and p.get("res") is None and not p.get("accumulate")
                                                       and os.environ.get("YL_FP8_OUT8_ONLY", "1") != "0")
                    if only_ok and all(cand(r) and key(r["inp"]) == key(x) and not (planes(r.get("res")) & planes(p["out"]))
                                       for r in ops if reads(r) & planes(p["out"])):
                        p["out8_only"] = 1
                else:
                    out.append(dict(kind=OP_QUANT, name=o["name"] + ".q8", inp=x, out=q, res=None, out_scale=sc,
                                    lane=o.get("lane", 0)))
                live[key(x)] = (q, sc)
            q, s_in = live[key(x)]
            o = dict(o, inp=q, in_scale=s_in)
            # all readers of this conv's output (until something overwrites it) are fp8 convs -> write e4m3 directly
            y = o["out"]
            rd, direct = [], True
            for r in ops[i + 1:]:
                if reads(r) & wrote:
                    rd.append(r)
                    if not (cand(r) and key(r["inp"]) == key(y) and not (planes(r.get("res")) & wrote)
                            and r.get("lane", 0) == o.get("lane", 0)):
                        direct = False
                if planes(r.get("out")) & wrote:
                    break
            if direct and rd:
                y8 = ar.alloc(y.H, y.W, y.C, FP8)
                s_out = float(scales[rd[0]["name"]])
                o["out"], o["out_scale"] = y8, s_out
                out.append(o)
                live[key(y)] = (y8, s_out)
                continue
        out.append(o)
        for k in [k for k in live if {(k[0], p) for p in range(k[1] // 8, (k[1] + k[2]) // 8)} & wrote]:
            del live[k]
    return out


def fold_conv_bn(sd: Dict[str, torch.Tensor], name: str, bn: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """BN-folded fp32 (weight OIHW, bias) of the Conv unit ``name`` (torch_utils.py:182-209).  A unit that already
    carries ``conv.bias`` and no BN tensors (``nn/export.py``'s folded form) is returned as stored."""
    if bn and name + ".bn.weight" not in sd and name + ".conv.bias" in sd:
        w = sd[name + ".conv.weight"].detach().float()
        bias = sd[name + ".conv.bias"].detach().float()
    elif bn:
        w = sd[name + ".conv.weight"].detach().float()
        g, b = sd[name + ".bn.weight"].float(), sd[name + ".bn.bias"].float()
        mu, var = sd[name + ".bn.running_mean"].float(), sd[name + ".bn.running_var"].float()
        s = g / torch.sqrt(var + BN_EPS)
        w = w * s.view(-1, 1, 1, 1)
        bias = b - mu * s
    else:
        w = sd[name + ".weight"].detach().float()
        bias = sd[name + ".bias"].detach().float()
    return np.ascontiguousarray(w.numpy(), dtype=np.float32), np.ascontiguousarray(bias.numpy(), dtype=np.float32)


def side_lane_head(ops: List[dict], det_prefix: str, feeds: Sequence[int], side_levels: int) -> List[dict]:
    """The one-batch profile's head placement (``lower(one_batch=True)``): the Detect branches of the first ``side_levels``
    levels go on SIDE LANE 1 right behind the layer that feeds them (``feeds[lvl]`` = its layer index), so that a caller who
    runs one batch at a time has the chip-filling 80x80 branches overlap the 40x40 / 20x20 neck, whose launches cannot fill
    256 CUs (head.py:59-62: the levels are independent; yolo11.yaml:33-46: the neck goes on from layer 16 while level 0's
    branches only need layer 16).  The engine runs a program with lanes as one graph per segment on a stream per lane
    (csrc/engine.hip detect_lanes).  The remaining levels' copies of each conv are grouped as ``group_detect_levels`` does
    and run behind the network.  Same kernels per conv as ungrouped launches: bit-identical."""
    head = [o for o in ops if o.get("name", "").startswith(det_prefix + ".")]
    nlv = len(feeds)
    if len(head) != 4 * nlv or any(o.get("lane", 0) for o in head) or not 0 < side_levels < nlv:
        return ops

    def level_ops(lvl: int) -> Optional[List[dict]]:
        names = (f"{det_prefix}.cv3.{lvl}.0.0+pw", f"{det_prefix}.cv3.{lvl}.1.0+pw+2", f"{det_prefix}.cv2.{lvl}.0", f"{det_prefix}.cv2.{lvl}.1+2")
        got = [o for nm in names for o in head if o["name"] == nm]
        return got if len(got) == 4 else None

    per_level = [level_ops(lvl) for lvl in range(nlv)]
    if any(v is None for v in per_level):
        return ops
    rest = [o for o in ops if not o.get("name", "").startswith(det_prefix + ".")]
    out: List[dict] = list(rest)
    for lvl in range(side_levels):
        pref = f"model.{feeds[lvl]}."
        last = max((i for i, o in enumerate(out) if o.get("name", "").startswith(pref) and o.get("lane", 0) == 0), default=None)
        if last is None:
            return ops
        for o in per_level[lvl]:
            o["lane"] = 1
        out[last + 1:last + 1] = per_level[lvl]
    tail_lv = list(range(side_levels, nlv))
    if len(tail_lv) >= 2:
        first_cls = [per_level[lvl][0] for lvl in tail_lv]
        box0 = [per_level[lvl][2] for lvl in tail_lv]
        box1 = [per_level[lvl][3] for lvl in tail_lv]
        cls1 = [per_level[lvl][1] for lvl in tail_lv]
        for o in box0:
            o["group"] = 1
        for o in box1 + cls1:
            o["group"] = 2
        out += first_cls + box0 + box1 + cls1
    else:
        for lvl in tail_lv:
            cl0, cl1, b0, b1 = per_level[lvl]
            out += [cl0, b0, b1, cl1]
    return out


def _out_hw(h: int, w: int, k: int, s: int) -> Tuple[int, int]:
    p = k // 2
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


def group_detect_levels(ops: List[dict], det_prefix: str, nlv: int, keep_first_box_conv: bool = False,
                        join_branches: bool = False, join_first_only: bool = False) -> List[dict]:
    """The Detect head runs the same two branches on every level (head.py:59-62, `for i in range(self.nl)`); the 40x40
    and 20x20 copies cannot fill the chip by themselves.  Moves the head to the end of the program, the levels' copies of
    each conv side by side, and marks the copies that share one tiling as a GROUP (``yl_op.group``): the engine runs a
    group as one launch (conv_group_kernel / dwpw_group_kernel).  Buffers are never recycled (``_Arena``), so moving an
    op later only needs its inputs to exist -- the head reads the neck's outputs and itself.  Same arithmetic per conv:
    results are bit-identical to separate launches (tests/test_gpu_net.py)."""
    head = [o for o in ops if o.get("name", "").startswith(det_prefix + ".")]
    if len(head) != 4 * nlv or any(o.get("lane", 0) for o in head):
        return ops  # not the fully fused form (box: conv, conv+1x1; class: dw+pw, dw+pw+1x1): leave the program alone

    def role(name: str, lvl: int):
        for r, pat in enumerate((f"{det_prefix}.cv3.{lvl}.0.0+pw", f"{det_prefix}.cv2.{lvl}.0", f"{det_prefix}.cv2.{lvl}.1+2",
                                 f"{det_prefix}.cv3.{lvl}.1.0+pw+2")):
            if name == pat:
                return r
        return None

    by_role: Dict[int, List[dict]] = {}
    for lvl in range(nlv):
        for o in head:
            r = role(o["name"], lvl)
            if r is not None:
                by_role.setdefault(r, []).append(o)
    if sorted(by_role) != [0, 1, 2, 3] or any(len(v) != nlv for v in by_role.values()):
        return ops
    rest = [o for o in ops if not o.get("name", "").startswith(det_prefix + ".")]
    if join_branches and nlv == 3:
        # The box and the class branch are independent too (head.py:61): their launches join as well (head_group_kernel).
        # Launch 1 = the first box conv of every level + the class branch's first half of levels 0 and 1 (level 2's has
        # 256 input channels: its LDS tile would be charged to every workgroup of the launch, so it runs alone, first);
        # launch 2 = the second box conv + 1x1 of every level + the class branch's second half of every level.
        for o in by_role[1] + by_role[0][:2]:
            o["group"] = 1
        if join_first_only:  # (the second halves each fill the chip by themselves: separate group launches)
            for o in by_role[2]:
                o["group"] = 2
            for o in by_role[3]:
                o["group"] = 3
        else:
            for o in by_role[2] + by_role[3]:
                o["group"] = 2
        return rest + by_role[0][2:] + by_role[1] + by_role[0][:2] + by_role[2] + by_role[3]
    if os.environ.get("YL_GROUP_CLS0", "0") == "1" and nlv >= 2:
        # the first halves of the class branch on levels 0 and 1 as one launch (level 2's 256 input channels would charge its
        # LDS tile to every workgroup: it keeps its own launch)
        for o in by_role[0][:2]:
            o["group"] = 4
    for r in (1, 2, 3):  # role 0 (first half of the class branch) differs in width per level: separate launches
        for lvl, o in enumerate(by_role[r]):
            if not (keep_first_box_conv and r == 1 and lvl == 0):  # (level 0's first box conv may keep its own kernel)
                o["group"] = r
    return rest + by_role[0] + by_role[1] + by_role[2] + by_role[3]


def lower(specs: Sequence[LayerSpec], sd: Dict[str, torch.Tensor], batch: int, in_h: int, in_w: int,
          strides: Sequence[float] = (8.0, 16.0, 32.0), side_lanes: bool = True, fuse_post: bool = True,
          fused_stem: bool = True, fp8_scales: Optional[Dict[str, float]] = None, one_batch: bool = False,
          dfl_epilogue: bool = True, keep_logits: bool = False) -> Program:
    """``dfl_epilogue`` (default on): Detect's box decode -- DFL + dist2bbox * stride (head.py:95-126, block.py:51-69, tal.py:341-350) --
    runs in the epilogue of the conv that computes a level's 64 DFL logits (cv2[i][2], alone or fused behind cv2[i][1]) on its fp32
    accumulators and stores (cx, cy, w, h) fp32 per anchor (``Program.levels[i]["offset_xywh"]``); the logits are stored as bf16
    planes too only with ``keep_logits`` (parity tests; ``levels[i]["offset"]`` is -1 otherwise).  ``dfl_epilogue=False``: the form of
    rounds 1-5 -- bf16 logits stored, decoded by the NMS / decode kernels (box error up to 0.3 stride from the bf16 rounding).

    ``one_batch=True``: the program of a caller that runs ONE batch at a time (``predict()`` per call): the resident C3k blocks take two
    workgroups per image (0.83 of the latency for 1.4 x the chip time; with batches in flight it costs 1.5 % of the throughput), and
    SPPF / the PSABlock stay separate launches (their one-launch forms win chip time, not latency), and the Detect levels' box and
    class branch launches are joined (YL_GROUP_LEVELS=3)."""
    det = specs[-1]
    assert det.kind == "Detect"
    prog = Program(batch, in_h, in_w, det.nc, strides=tuple(float(s) for s in strides))
    ar = _Arena(batch)
    ops = prog.ops

    # ---- shapes of every layer output
    shp: List[Tuple[int, int, int]] = []
    for sp in specs[:-1]:
        if isinstance(sp.f, int):
            h, w, _ = (in_h, in_w, 3) if (sp.i == 0) else shp[sp.f if sp.f != -1 else sp.i - 1]
        else:
            h, w, _ = shp[sp.f[0] if sp.f[0] != -1 else sp.i - 1]
        if sp.kind == "Conv":
            h, w = _out_hw(h, w, sp.k, sp.s)
        elif sp.kind == "Upsample":
            h, w = 2 * h, 2 * w
        shp.append((h, w, sp.c2))

    # ---- Concat placement: each member gets a channel slice of the concat buffer
    dest: Dict[int, TView] = {}
    cat_buf: Dict[int, TView] = {}
    late_copy: Dict[int, List[Tuple[int, TView]]] = {}
    for sp in specs:
        if sp.kind != "Concat":
            continue
        h, w, c = shp[sp.i]
        buf = ar.alloc(h, w, c)
        cat_buf[sp.i] = buf
        coff = 0
        for j, cj in zip(sp.f, sp.c1):
            src = sp.i - 1 if j == -1 else j
            sl = buf.slice(coff, cj)
            if src in dest or specs[src].kind == "Concat":
                late_copy.setdefault(sp.i, []).append((src, sl))
            else:
                dest[src] = sl
            coff += cj

    def out_view(i: int) -> TView:
        if i in dest:
            return dest[i]
        h, w, c = shp[i]
        return ar.alloc(h, w, c)

    def conv(name, x: TView, y: TView, k=1, s=1, act=True, res: Optional[TView] = None, bn=True,
             wb: Optional[Tuple[np.ndarray, np.ndarray]] = None, cin_weight=0):
        w, b = wb if wb is not None else fold_conv_bn(sd, name, bn)
        assert w.shape[0] == y.C and w.shape[2] == k, (name, w.shape, y.C)
        assert (cin_weight or x.C) == w.shape[1], (name, w.shape, x.C)
        ops.append(dict(kind=OP_CONV, name=name, k=k, stride=s, act=int(act), inp=x, out=y, res=res,
                        weight=w, bias=b, cin_weight=cin_weight))

    def dwconv(name, x: TView, y: TView, act=True, accumulate=False, grp=0, grp_stride=0, grp_off=0):
        w, b = fold_conv_bn(sd, name, True)
        assert w.shape[1] == 1 and w.shape[0] == y.C, (name, w.shape)
        ops.append(dict(kind=OP_DWCONV3, name=name, k=3, stride=1, act=int(act), accumulate=int(accumulate),
                        grp=grp, grp_stride=grp_stride, grp_off=grp_off, inp=x, out=y, res=None, weight=w, bias=b))

    def bottleneck(p, x: TView, y: TView, c_mid: int, in_place_res: bool = True):
        tmp = ar.alloc(x.H, x.W, c_mid)
        conv(p + ".cv1", x, tmp, 3, 1)
        conv(p + ".cv2", tmp, y, 3, 1, res=x if x.C == y.C else None)

    def c3k_block(p, x: TView, y: TView):
        c = x.C
        c_ = int(c * 0.5)
        u = ar.alloc(x.H, x.W, 2 * c_)
        w1, b1 = fold_conv_bn(sd, p + ".cv1")
        w2, b2 = fold_conv_bn(sd, p + ".cv2")
        # the whole block in ONE launch, one workgroup per image, every tensor between x and y in LDS / registers
        # (csrc/resident.hip; YOLO11n layers 8 and 22 at 640x640: 6 launches -> 1, bit-identical).  A workgroup per image
        # fills the chip only with a batch behind it: small batches keep the six launches (YL_FUSE_C3K_MIN_BATCH).
        if (fuse_post and os.environ.get("YL_FUSE_C3K", "1") != "0" and c3k_resident_fits(c_, x.H, x.W)
                and c == 2 * c_ and y.C == c and batch >= int(os.environ.get("YL_FUSE_C3K_MIN_BATCH", "8")) and not fp8_scales):
            w3, b3 = fold_conv_bn(sd, p + ".cv3")
            xw, xb, names = [], [], [p + ".cv1+cv2"]
            for j in range(2):
                for q in ("cv1", "cv2"):
                    wj, bj = fold_conv_bn(sd, f"{p}.m.{j}.{q}")
                    xw.append(wj)
                    xb.append(bj)
                    names.append(f"{p}.m.{j}.{q}")
            xw.append(w3.reshape(w3.shape[0], -1))
            xb.append(b3)
            names.append(p + ".cv3")
            ops.append(dict(kind=OP_C3K, name=p, act=1, inp=x, out=y, res=None, split=2 if one_batch else 0,
                            weight=np.concatenate([w1, w2], 0).reshape(2 * c_, -1), bias=np.concatenate([b1, b2], 0),
                            extra_weight=xw, extra_bias=xb, fused_names=names))
            return
        # C3.cv1 and C3.cv2 read the same input (block.py:259): one conv with stacked filters
        conv(p + ".cv1+cv2", x, u, 1, 1, wb=(np.concatenate([w1, w2], 0), np.concatenate([b1, b2], 0)))
        a = u.slice(0, c_)
        fb = os.environ.get("YL_FUSE_BNECK", "1")
        if fuse_post and fb != "0" and c_ in ((32, 64) if fb == "2" else (32,)):
            # each Bottleneck (3x3, 3x3, + x) in ONE launch (csrc/bneck2.hip).  A workgroup reads a halo of its
            # input, so the in-place update of the unfused form would race: the two Bottlenecks ping-pong between
            # the block's buffer and a scratch buffer instead.  Default: c = 32 only (40x40 in YOLO11n: 15 us against
            # 19 us for the two launches); c = 64 (20x20: 160 workgroups of 4 x 32 outputs, 25 us against 24 us) is
            # opt-in with YL_FUSE_BNECK=2.
            t2 = ar.alloc(x.H, x.W, c_)
            for j, (src_v, dst_v) in enumerate(((a, t2), (t2, a))):
                w1_, b1_ = fold_conv_bn(sd, f"{p}.m.{j}.cv1")
                w2_, b2_ = fold_conv_bn(sd, f"{p}.m.{j}.cv2")
                ops.append(dict(kind=OP_BNECK2, name=f"{p}.m.{j}", act=1, inp=src_v, out=dst_v, res=None, weight=w1_, bias=b1_,
                                post_weight=w2_, post_bias=b2_, fused_names=[f"{p}.m.{j}.cv2"]))
        else:
            for j in range(2):
                bottleneck(f"{p}.m.{j}", a, a, c_)
        conv(p + ".cv3", u, y, 1, 1)

    # ---- Detect (head.py:29-69): box branch cv2, class branch cv3, raw level maps as bf16 planes.
    # Each level's two branches are emitted right after the layer that feeds them, on side lanes:
    # they only depend on that layer, so the engine runs them concurrently with the rest of the
    # neck (whose 20x20 / 40x40 layers cannot fill the chip by themselves).
    dp = f"model.{det.i}"
    c2, c3 = detect_mid_channels(det.c1, det.nc)
    level_of = {j: lvl for lvl, j in enumerate(det.f)}
    level_rec: Dict[int, dict] = {}

    # Lane policy.  side_lanes=True: the class branch of the first (largest) level runs on side lane 1,
    # everything else on the main stream (measured best on MI355X among ~15 maps: cross-stream edges
    # inside a hipGraph cost 5-60 us each, so only the longest chain -- five 80x80 kernels that overlap
    # the 80x80 box branch and the 40x40 neck -- is worth a fork; 1.37 -> 1.21 ms end to end).
    # YL_LANES="b0c0b1c1b2c2" (one lane digit per level and branch) overrides it for experiments.
    # Round 2: OFF by default (YL_SIDE_LANES=1 restores it).  Whether a captured graph's branch really overlaps the trunk is
    # up to the HIP runtime, and the one setting that made it do so reliably (DEBUG_HIP_FORCE_GRAPH_QUEUES=2) replays the
    # graph WITHOUT honouring the fork edge (see _lib.RUNTIME_KNOBS): the hot path is captured as one chain, correctness
    # does not depend on the runtime's multi-queue graph scheduling, and overlap comes from batches in flight.
    import os
    side_lanes = side_lanes and os.environ.get("YL_SIDE_LANES", "0") == "1"
    nlv = len(det.f)
    lanes = [0] * (2 * nlv)
    if side_lanes and nlv > 1:
        lanes[1] = 1
    env = os.environ.get("YL_LANES", "")
    if len(env) == 2 * nlv and env.isdigit():
        lanes = [int(ch) for ch in env]

    def box_lane(lvl: int) -> int:
        return lanes[2 * lvl]

    def cls_lane(lvl: int) -> int:
        return lanes[2 * lvl + 1]

    # Emission order = issue order on the main stream.  With side lanes the (large, bandwidth-heavy) box
    # branch of level 0 is issued after the 20x20 / 40x40 tail of the network: the side lane's large
    # class-branch kernels then overlap the small latency-bound kernels instead of the other large branch
    # (measured 1.124 -> 1.077 ms end to end; pairing two large branches makes both slower).
    # YL_BOX_DEFER="<level digits>" overrides the set of deferred box branches ("-" = none).
    deferred: List[dict] = []
    defer_box = {0} if (side_lanes and nlv > 1) else set()
    if "YL_BOX_DEFER" in os.environ:
        defer_box = {int(ch) for ch in os.environ["YL_BOX_DEFER"] if ch.isdigit()}

    def emit_level(lvl: int, xin: TView):
        h, w = xin.H, xin.W
        # raw level maps: ordinary bf16 planar buffers, 64 DFL logits and nc8 class logits (nc padded to a
        # multiple of 8 with zero filters and a -1e4 bias: such a channel can never be the best class)
        nc8 = (det.nc + 7) // 8 * 8
        head_box = ar.alloc(h, w, 64)
        head_cls = ar.alloc(h, w, nc8)
        n0 = len(ops)
        t1, t2 = ar.alloc(h, w, c2), ar.alloc(h, w, c2)
        conv(f"{dp}.cv2.{lvl}.0", xin, t1, 3, 1)
        conv(f"{dp}.cv2.{lvl}.1", t1, t2, 3, 1)
        conv(f"{dp}.cv2.{lvl}.2", t2, head_box, 1, 1, act=False, bn=False)
        off_xywh = -1
        if dfl_epilogue:
            off_xywh = ar.alloc_bytes(batch * h * w * 16)  # fp32 (cx, cy, w, h) per anchor, [B][H * W][4]
            ops[-1]["dfl_level"], ops[-1]["dfl_keep"] = lvl + 1, int(keep_logits)
        for o in ops[n0:]:
            o["lane"] = box_lane(lvl)
        if lvl in defer_box:  # run this box branch after the rest of the network (see below)
            deferred.extend(ops[n0:])
            del ops[n0:]
        n1 = len(ops)
        d1, e1 = ar.alloc(h, w, xin.C), ar.alloc(h, w, c3)
        dwconv(f"{dp}.cv3.{lvl}.0.0", xin, d1)
        conv(f"{dp}.cv3.{lvl}.0.1", d1, e1, 1, 1)
        d2, e2 = ar.alloc(h, w, c3), ar.alloc(h, w, c3)
        dwconv(f"{dp}.cv3.{lvl}.1.0", e1, d2)
        conv(f"{dp}.cv3.{lvl}.1.1", d2, e2, 1, 1)
        wc, bc = fold_conv_bn(sd, f"{dp}.cv3.{lvl}.2", bn=False)
        if nc8 != det.nc:
            wc = np.concatenate([wc, np.zeros((nc8 - det.nc, *wc.shape[1:]), np.float32)], 0)
            bc = np.concatenate([bc, np.full(nc8 - det.nc, -1e4, np.float32)], 0)
        conv(f"{dp}.cv3.{lvl}.2", e2, head_cls, 1, 1, act=False, wb=(wc, bc))
        for o in ops[n1:]:
            o["lane"] = cls_lane(lvl)
        level_rec[lvl] = dict(offset=head_box.offset if (keep_logits or not dfl_epilogue) else -1, offset_cls=head_cls.offset,
                              offset_xywh=off_xywh, nc8=nc8, H=h, W=w, stride=float(strides[lvl]))

    views: Dict[int, TView] = {}
    sppf_tail: Dict[int, TView] = {}  # C2PSA layer -> its cat(a, b) buffer, when the SPPF launch in front of it ran its cv1
    prog.input_view = ar.alloc(in_h, in_w, 8)
    x = prog.input_view
    for sp in specs[:-1]:
        p = f"model.{sp.i}"
        src = None
        if isinstance(sp.f, int):
            src = x if sp.f == -1 else views[sp.f]
        if sp.kind == "Conv" and sp.i == 0 and sp.c1 == 3 and sp.k == 3 and sp.s == 2 and sp.c2 % 16 == 0 and fused_stem:
            # layer 0 reads the NCHW network input directly (no converted NHWC copy of the input)
            y = out_view(sp.i)
            w, b = fold_conv_bn(sd, p)
            ops.append(dict(kind=OP_STEM, name=p, k=3, stride=2, act=1, inp=src, out=y, res=None, weight=w, bias=b))
        elif sp.kind == "Conv":
            y = out_view(sp.i)
            conv(p, src, y, sp.k, sp.s, cin_weight=3 if sp.i == 0 and src.C == 8 and sp.c1 == 3 else 0)
        elif sp.kind == "C3k2":
            y = out_view(sp.i)
            c = int(sp.c2 * sp.e)
            t = ar.alloc(src.H, src.W, (2 + sp.n) * c)
            conv(p + ".cv1", src, t.slice(0, 2 * c), 1, 1)
            for j in range(sp.n):
                xin, xout = t.slice((1 + j) * c, c), t.slice((2 + j) * c, c)
                if sp.c3k:
                    c3k_block(f"{p}.m.{j}", xin, xout)
                else:
                    bottleneck(f"{p}.m.{j}", xin, xout, int(c * 0.5))
            conv(p + ".cv2", t, y, 1, 1)
        elif sp.kind == "SPPF":
            c_ = sp.c1 // 2
            # the whole layer in ONE launch, one or two workgroups per image (csrc/sppf_resident.hip: x and the pooled maps in
            # LDS) -- and C2PSA.cv1 with it when that conv is the only reader of this layer's output (YOLO11: layer 9 -> 10;
            # 4 launches -> 1, bit-identical); small batches keep the separate launches, like the other resident kernels
            nxt = specs[sp.i + 1] if sp.i + 1 < len(specs) - 1 else None
            only_reader = (nxt is not None and nxt.kind == "C2PSA" and nxt.f == -1 and sp.i not in dest and sp.i not in level_of
                           and not any(sp.i in (q.f if isinstance(q.f, (list, tuple)) else [q.f]) for q in specs))
            tail_c = 2 * int(nxt.c1 * 0.5) if only_reader else 0
            # (not in the one-batch profile: 55.6 us against 52.5 us in sequence -- it pays in chip time, not in latency)
            if (fuse_post and os.environ.get("YL_FUSE_SPPF", "1") != "0" and not fp8_scales and not one_batch
                    and batch >= int(os.environ.get("YL_FUSE_C3K_MIN_BATCH", "8"))
                    and (sppf_resident_fits(sp.c1, sp.c2, tail_c, src.H, src.W) or sppf_resident_fits(sp.c1, sp.c2, 0, src.H, src.W))):
                if not sppf_resident_fits(sp.c1, sp.c2, tail_c, src.H, src.W):
                    tail_c = 0
                w1, b1 = fold_conv_bn(sd, p + ".cv1")
                w2, b2 = fold_conv_bn(sd, p + ".cv2")
                xw, xb, names = [w2.reshape(w2.shape[0], -1)], [b2], [p + ".cv1", p + ".m", p + ".cv2"]
                if tail_c:
                    w3, b3 = fold_conv_bn(sd, f"model.{nxt.i}.cv1")
                    xw.append(w3.reshape(w3.shape[0], -1))
                    xb.append(b3)
                    names.append(f"model.{nxt.i}.cv1")
                    y = ar.alloc(src.H, src.W, tail_c)  # = C2PSA's cat(a, b) buffer; this layer's own output is never stored
                    sppf_tail[nxt.i] = y
                else:
                    y = out_view(sp.i)
                ops.append(dict(kind=OP_SPPF, name=p, act=1, inp=src, out=y, res=None, weight=w1.reshape(w1.shape[0], -1), bias=b1,
                                extra_weight=xw, extra_bias=xb, fused_names=names))
            else:
                y = out_view(sp.i)
                sbuf = ar.alloc(src.H, src.W, 4 * c_)
                conv(p + ".cv1", src, sbuf.slice(0, c_), 1, 1)
                ops.append(dict(kind=OP_SPPF_POOL, name=p + ".m", inp=sbuf.slice(0, c_), out=sbuf.slice(0, c_), res=None))
                conv(p + ".cv2", sbuf, y, 1, 1)
        elif sp.kind == "C2PSA":
            y = out_view(sp.i)
            c = int(sp.c1 * 0.5)
            nh = c // 64
            hd = c // nh
            kd = int(hd * 0.5)
            if sp.i in sppf_tail:  # cv1 ran in the SPPF layer's launch: `src` already is its output
                u = sppf_tail[sp.i]
            else:
                u = ar.alloc(src.H, src.W, 2 * c)
                conv(p + ".cv1", src, u, 1, 1)
            b = u.slice(c, c)
            psa_tail_done = False
            for j in range(sp.n):
                q = f"{p}.m.{j}"
                # the whole PSABlock in ONE launch, one workgroup per image (csrc/resident.hip psa_resident_kernel: 6 launches ->
                # 1, bit-identical); small batches keep the six launches, like the resident C3k block
                # (the one-batch profile keeps the six launches: 69.8 us in sequence against 71 us; with FOUR workgroups per image --
                # YL_LAT_PSA=1 -- the one-launch form measures the same there, 34.4 against 34.5 k one batch at a time)
                if (fuse_post and os.environ.get("YL_FUSE_PSA", "1") != "0" and not (one_batch and os.environ.get("YL_LAT_PSA", "0") != "1")
                        and psa_resident_fits(c, nh, kd, hd, src.H, src.W)
                        and batch >= int(os.environ.get("YL_FUSE_C3K_MIN_BATCH", "8")) and not fp8_scales):
                    wq, bq = fold_conv_bn(sd, q + ".attn.qkv")
                    names, xw, xb = [q + ".attn.qkv", q + ".attn"], [], []
                    for part in (".attn.pe", ".attn.proj", ".ffn.0", ".ffn.1"):
                        wj, bj = fold_conv_bn(sd, q + part)
                        xw.append(wj.reshape(wj.shape[0], -1) if wj.shape[2] == 1 else wj)
                        xb.append(bj)
                        names.append(q + part)
                    # (in -> out, never in place: cat(a, b) of this block's input buffer becomes cat(a, b') in a fresh one)
                    u2 = ar.alloc(src.H, src.W, 2 * c)
                    ops.append(dict(kind=OP_PSA, name=q, act=0, heads=nh, key_dim=kd, head_dim=hd, scale=float(kd ** -0.5), split=4 if one_batch else 0,
                                    inp=u, out=u2, res=None, weight=wq.reshape(wq.shape[0], -1), bias=bq, extra_weight=xw,
                                    extra_bias=xb, fused_names=names))
                    u = u2
                    b = u.slice(c, c)
                    continue
                qkv = ar.alloc(src.H, src.W, c + 2 * nh * kd)
                conv(q + ".attn.qkv", b, qkv, 1, 1, act=False)
                o = ar.alloc(src.H, src.W, c)
                ops.append(dict(kind=OP_ATTENTION, name=q + ".attn", heads=nh, key_dim=kd, head_dim=hd,
                                scale=float(kd ** -0.5), inp=qkv, out=o, res=None))
                dwconv(q + ".attn.pe", qkv, o, act=False, accumulate=True, grp=hd, grp_stride=2 * kd + hd,
                       grp_off=2 * kd)
                # everything behind the attention is pointwise: proj, ffn.0, ffn.1 and C2PSA.cv2 in ONE launch over the pixels
                # of the whole batch (csrc/psa_tail.hip, bit-identical), where the one-launch PSABlock is not taken -- i.e. in the
                # one-batch profile: four launch skeletons (42 us in sequence at batch 32) for 0.9 us of matrix work
                if (fuse_post and os.environ.get("YL_FUSE_PSA_TAIL", "1") != "0" and psa_tail_fits(c, y.C, sp.n)
                        and u.coff == 0 and not fp8_scales):
                    names, xw, xb = [q + ".attn.proj"], [], []
                    wp, bpj = fold_conv_bn(sd, q + ".attn.proj")
                    for full in (q + ".ffn.0", q + ".ffn.1", p + ".cv2"):
                        wj, bj = fold_conv_bn(sd, full)
                        xw.append(wj.reshape(wj.shape[0], -1))
                        xb.append(bj)
                        names.append(full)
                    ops.append(dict(kind=OP_PSA_TAIL, name=q + ".tail", act=1, inp=o, out=y, res=u, weight=wp.reshape(wp.shape[0], -1),
                                    bias=bpj, extra_weight=xw, extra_bias=xb, fused_names=names))
                    psa_tail_done = True
                    continue
                conv(q + ".attn.proj", o, b, 1, 1, act=False, res=b)
                f1 = ar.alloc(src.H, src.W, 2 * c)
                conv(q + ".ffn.0", b, f1, 1, 1)
                conv(q + ".ffn.1", f1, b, 1, 1, act=False, res=b)
            if not psa_tail_done:
                conv(p + ".cv2", u, y, 1, 1)
        elif sp.kind == "Upsample":
            y = out_view(sp.i)
            ops.append(dict(kind=OP_UPSAMPLE, name=p, stride=2, inp=src, out=y, res=None))
        elif sp.kind == "Concat":
            y = cat_buf[sp.i]
            for s_i, sl in late_copy.get(sp.i, []):
                ops.append(dict(kind=OP_UPSAMPLE, name=f"{p}.copy{s_i}", stride=1, inp=views[s_i], out=sl, res=None))
        else:
            raise NotImplementedError(sp.kind)
        views[sp.i] = y
        x = y
        if sp.i in level_of:
            emit_level(level_of[sp.i], y)

    ops.extend(deferred)
    if fuse_post and os.environ.get("YL_FUSE_POST", "1") != "0":
        ops[:] = fuse_follow_up_convs(ops)
    if fuse_post and os.environ.get("YL_FUSE_DWPW", "1") != "0":
        ops[:] = fuse_depthwise_pointwise(ops)
        # Detect's last class conv joins the fused depthwise + pointwise launch (bit-identical).  Round 1 measured this
        # SLOWER end to end (34.7-37.1 k vs 40.9-41.7 k images/s) because the 80x80 class branch ran on the graph's side
        # lane beside the trunk; with the graph captured as one chain (round 2) it is +2.5 % (43.9 k -> 45.0 k images/s,
        # 72 -> 69 launches).  YL_FUSE_DWPW_TAIL=0 restores the separate 1x1 launches.
        if os.environ.get("YL_FUSE_DWPW_TAIL", "1") == "1":
            ops[:] = fuse_dwpw_tail(ops)
            # The Detect levels' copies of one conv in one launch (round 2): 69 -> 64 launches, one batch at a time 1.106 ->
            # 1.06 ms.  YL_GROUP_LEVELS=2 (default) leaves level 0's first box conv on its own kernel (conv3_reg_kernel);
            # =1 groups it too (one launch fewer, -1 % with 20 batches in flight); =0 separate launches.
            # (the latency profile joins the box and class branch launches of the levels too, =3: 42 -> 38 launches, one batch at a
            # time +2.4 %, with batches in flight -2.7 %)
            gl = os.environ.get("YL_GROUP_LEVELS", "3" if one_batch else "2")
            # (one batch at a time: the first levels' branches on a side lane, see side_lane_head; YL_ONE_BATCH_SIDE = levels
            # that go there, 0 = none: the joined group launches of round 4)
            # MEASURED (round 5, NOTEBOOK section 15) and OFF by default: the branches do overlap the neck (kernel trace), but the
            # neck's small launches slow down 1.8 x beside them and the remaining levels lose the three-level group launches:
            # 907-913 us per batch against 895-901 us.
            side_n = int(os.environ.get("YL_ONE_BATCH_SIDE", "0")) if one_batch and not fp8_scales else 0
            if side_n > 0 and "YL_GROUP_LEVELS" not in os.environ:
                n_before = len(ops)
                ops[:] = side_lane_head(ops, dp, list(det.f), side_n)
                assert len(ops) == n_before
                if any(o.get("lane", 0) for o in ops):
                    gl = "done"
            if gl in ("1", "2", "3", "4") and not fp8_scales:
                ops[:] = group_detect_levels(ops, dp, nlv, keep_first_box_conv=gl == "2", join_branches=gl in ("3", "4"),
                                             join_first_only=gl == "4")
    if fuse_post and os.environ.get("YL_FUSE_UP", "1") != "0":
        ops[:] = fuse_upsample_into_convs(ops)
    if fuse_post and os.environ.get("YL_FUSE_STEM", "1") != "0":
        ops[:] = fuse_stem_into_conv(ops, in_w)
    if fuse_post and os.environ.get("YL_FUSE_C3K2", "1") != "0":
        ops[:] = fuse_c3k2_small(ops)
    if one_batch and os.environ.get("YL_LAT_CACHEABLE", "1") != "0":
        # one batch at a time: the readers of the fused stem's / the narrow C3k2 blocks' outputs are the next launches -- keep those
        # outputs cacheable (no non-temporal hint on the stores; yl_op.reserved3 bit 8): layer 2 then takes 45.5 instead of 54.1 us
        for o in ops:
            if o["kind"] == OP_C3K2S or (o["kind"] == OP_CONV and "stem_weight" in o):
                o["cacheable"] = 1
    if fp8_scales:
        # opt-in: e4m3 mirrors make the 1x1 convs fp8 too (90 % of YOLO11x's conv FLOPs; 923 -> 1023 images/s at 1280x1280
        # batch 8) but the detections then leave the stated fp8 tolerance
        if os.environ.get("YL_FP8_MIRROR", "0") == "1":
            ops[:] = apply_fp8_mirrors(ops, ar, fp8_scales)
        ops[:] = apply_fp8(ops, ar, fp8_scales)
    drop = [t for t in os.environ.get("YL_DROP_OPS", "").split(",") if t]  # timing ablation only: results are wrong
    if drop:
        ops[:] = [o for o in ops if not any(o.get("name", "").startswith(t) for t in drop)]
    a_total = 0
    for lvl in range(len(det.f)):
        prog.levels.append(level_rec[lvl])
        a_total += level_rec[lvl]["H"] * level_rec[lvl]["W"]
    prog.anchors = a_total
    prog.layer_views = views
    prog.ws_bytes = ar.top + 256
    return prog
