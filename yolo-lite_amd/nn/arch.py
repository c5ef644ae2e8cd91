"""YOLO11 architecture table and its channel/repeat arithmetic.

Mirrors what the reference derives in ``yololite/nn/tasks.py:525-664`` (``parse_model``) from
``yololite/cfg/yolo11.yaml:5-47`` and ``yaml_model_load`` (``tasks.py:667-680``), but produces a
plain list of :class:`LayerSpec` records (no ``nn.Module``): the HIP engine is lowered from these.
The same parser accepts the ``yaml`` dict pickled inside a ``.pt`` checkpoint.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

# depth, width, max_channels per scale letter (yolo11.yaml:7-12)
YOLO11_SCALES: Dict[str, Tuple[float, float, int]] = {
    "n": (0.50, 0.25, 1024),
    "s": (0.50, 0.50, 1024),
    "m": (0.50, 1.00, 512),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.50, 512),
}


def yolo11_dict(scale: str = "n", nc: int = 80) -> dict:
    """The yolo11 model dictionary in the same schema the reference's yaml loader returns."""
    up = "nn.Upsample"
    return {
        "nc": nc,
        "scale": scale,
        "scales": {k: list(v) for k, v in YOLO11_SCALES.items()},
        "backbone": [
            [-1, 1, "Conv", [64, 3, 2]],
            [-1, 1, "Conv", [128, 3, 2]],
            [-1, 2, "C3k2", [256, False, 0.25]],
            [-1, 1, "Conv", [256, 3, 2]],
            [-1, 2, "C3k2", [512, False, 0.25]],
            [-1, 1, "Conv", [512, 3, 2]],
            [-1, 2, "C3k2", [512, True]],
            [-1, 1, "Conv", [1024, 3, 2]],
            [-1, 2, "C3k2", [1024, True]],
            [-1, 1, "SPPF", [1024, 5]],
            [-1, 2, "C2PSA", [1024]],
        ],
        "head": [
            [-1, 1, up, [None, 2, "nearest"]],
            [[-1, 6], 1, "Concat", [1]],
            [-1, 2, "C3k2", [512, False]],
            [-1, 1, up, [None, 2, "nearest"]],
            [[-1, 4], 1, "Concat", [1]],
            [-1, 2, "C3k2", [256, False]],
            [-1, 1, "Conv", [256, 3, 2]],
            [[-1, 13], 1, "Concat", [1]],
            [-1, 2, "C3k2", [512, False]],
            [-1, 1, "Conv", [512, 3, 2]],
            [[-1, 10], 1, "Concat", [1]],
            [-1, 2, "C3k2", [1024, True]],
            [[16, 19, 22], 1, "Detect", ["nc"]],
        ],
    }


def make_divisible(x: float, divisor: int) -> int:
    """Smallest multiple of ``divisor`` that is >= x (reference ``utils/ops.py:101-114``)."""
    return int(math.ceil(x / divisor) * divisor)


def guess_model_scale(path: Union[str, Path]) -> str:
    """Scale letter from a file name such as ``yolo11n.yaml`` (reference ``tasks.py:683-700``)."""
    m = re.search(r"yolo[v]?\d+([nslmx])", Path(path).stem)
    return m.group(1) if m else ""


def yaml_model_load(path: Union[str, Path]) -> dict:
    """``yolo11{n,s,m,l,x}.yaml`` -> model dict with ``scale`` set (reference ``tasks.py:667-680``).

    A file that exists on disk is read with PyYAML; the built-in ``yolo11*.yaml`` names resolve to
    :func:`yolo11_dict` so no data file is needed.
    """
    path = Path(path)
    scale = guess_model_scale(path)
    unified = Path(re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", str(path)))
    for cand in (path, unified):
        if cand.is_file():
            import yaml

            with open(cand, errors="ignore", encoding="utf-8") as f:
                d = yaml.safe_load(f)
            d["scale"] = scale
            d["yaml_file"] = str(path)
            return d
    if re.fullmatch(r"yolo11[nslmx]?", path.stem):
        d = yolo11_dict(scale or "n")
        d["yaml_file"] = str(path)
        return d
    raise FileNotFoundError(f"model config '{path}' not found")


@dataclass
class LayerSpec:
    i: int
    f: Union[int, List[int]]
    kind: str  # Conv | C3k2 | SPPF | C2PSA | Upsample | Concat | Detect
    c1: Union[int, List[int]]
    c2: Optional[int]
    n: int = 1
    k: int = 1
    s: int = 1
    c3k: bool = False
    e: float = 0.5
    nc: int = 80
    extra: dict = field(default_factory=dict)


def parse_model(d: dict, ch: int = 3) -> Tuple[List[LayerSpec], List[int]]:
    """Model dict -> (layer specs, save list).  Follows reference ``tasks.py:525-664`` for the
    module kinds yolo11 uses; any other module name raises ``NotImplementedError``."""
    nc = d.get("nc", 80)
    scales = d.get("scales")
    depth, width, max_ch = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0), float("inf")
    scale = d.get("scale") or ""
    if scales:
        if not scale:
            scale = tuple(scales.keys())[0]
        depth, width, max_ch = scales[scale]
    chs: List[int] = []
    specs: List[LayerSpec] = []
    save: List[int] = []
    c2 = ch
    for i, (f, n, m, args) in enumerate(list(d["backbone"]) + list(d["head"])):
        args = [nc if a == "nc" else a for a in args]
        n = max(round(n * depth), 1) if n > 1 else n
        prev = lambda j: (ch if not chs else chs[-1]) if j == -1 else chs[j]  # noqa: E731
        if m in ("Conv", "C3k2", "SPPF", "C2PSA"):
            c1 = prev(f)
            c2 = args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_ch) * width, 8)
            sp = LayerSpec(i, f, m, c1, c2)
            if m == "Conv":
                sp.k = args[1] if len(args) > 1 else 1
                sp.s = args[2] if len(args) > 2 else 1
            elif m == "C3k2":
                sp.n = n
                sp.c3k = bool(args[1]) if len(args) > 1 else False
                sp.e = float(args[2]) if len(args) > 2 else 0.5
                if scale and scale in "mlx":
                    sp.c3k = True
            elif m == "C2PSA":
                sp.n = n
            elif m == "SPPF":
                sp.k = args[1] if len(args) > 1 else 5
        elif m in ("nn.Upsample", "Upsample"):
            c1 = prev(f)
            c2 = c1
            if (args[1] if len(args) > 1 else 2) != 2 or (len(args) > 2 and args[2] != "nearest"):
                raise NotImplementedError("only nearest 2x Upsample is on the yolo11 path")
            sp = LayerSpec(i, f, "Upsample", c1, c2)
        elif m == "Concat":
            c1 = [prev(j) for j in f]
            c2 = sum(c1)
            sp = LayerSpec(i, list(f), "Concat", c1, c2)
        elif m == "Detect":
            c1 = [prev(j) for j in f]
            sp = LayerSpec(i, list(f), "Detect", c1, None, nc=args[0] if args else nc)
            c2 = None
        else:
            raise NotImplementedError(f"module '{m}' is not part of the YOLO11 detect path")
        specs.append(sp)
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        chs.append(c2)
    return specs, sorted(set(save))


@dataclass
class ConvUnit:
    """One Conv2d (+BN +SiLU) of the graph: state-dict prefix and geometry."""
    name: str
    c1: int
    c2: int
    k: int = 1
    s: int = 1
    g: int = 1
    act: bool = True
    bn: bool = True  # False: plain nn.Conv2d with bias (Detect 1x1 heads)


def _c3k_units(p: str, c: int) -> List[ConvUnit]:
    c_ = int(c * 0.5)
    u = [ConvUnit(p + ".cv1", c, c_, 1), ConvUnit(p + ".cv2", c, c_, 1), ConvUnit(p + ".cv3", 2 * c_, c, 1)]
    for j in range(2):
        u += [ConvUnit(f"{p}.m.{j}.cv1", c_, c_, 3), ConvUnit(f"{p}.m.{j}.cv2", c_, c_, 3)]
    return u


def detect_mid_channels(ch: Sequence[int], nc: int) -> Tuple[int, int]:
    """(c2, c3) of ``Detect.__init__`` (reference ``head.py:36``)."""
    return max(16, ch[0] // 4, 64), max(ch[0], min(nc, 100))


def conv_units(specs: Sequence[LayerSpec]) -> List[ConvUnit]:
    """Every Conv2d of the model in state-dict order (reference module constructors:
    ``conv.py:35-53``, ``block.py:165-184,220-259,330-343,720-739,863-953,999-1038``, ``head.py:29-57``)."""
    out: List[ConvUnit] = []
    for sp in specs:
        p = f"model.{sp.i}"
        if sp.kind == "Conv":
            out.append(ConvUnit(p, sp.c1, sp.c2, sp.k, sp.s))
        elif sp.kind == "C3k2":
            c = int(sp.c2 * sp.e)
            out.append(ConvUnit(p + ".cv1", sp.c1, 2 * c, 1))
            out.append(ConvUnit(p + ".cv2", (2 + sp.n) * c, sp.c2, 1))
            for j in range(sp.n):
                q = f"{p}.m.{j}"
                if sp.c3k:
                    out += _c3k_units(q, c)
                else:
                    c_ = int(c * 0.5)
                    out += [ConvUnit(q + ".cv1", c, c_, 3), ConvUnit(q + ".cv2", c_, c, 3)]
        elif sp.kind == "SPPF":
            c_ = sp.c1 // 2
            out += [ConvUnit(p + ".cv1", sp.c1, c_, 1), ConvUnit(p + ".cv2", 4 * c_, sp.c2, 1)]
        elif sp.kind == "C2PSA":
            c = int(sp.c1 * 0.5)
            out += [ConvUnit(p + ".cv1", sp.c1, 2 * c, 1), ConvUnit(p + ".cv2", 2 * c, sp.c1, 1)]
            nh = c // 64
            hd = c // nh
            kd = int(hd * 0.5)
            for j in range(sp.n):
                q = f"{p}.m.{j}"
                out += [
                    ConvUnit(q + ".attn.qkv", c, c + 2 * nh * kd, 1, act=False),
                    ConvUnit(q + ".attn.proj", c, c, 1, act=False),
                    ConvUnit(q + ".attn.pe", c, c, 3, g=c, act=False),
                    ConvUnit(q + ".ffn.0", c, 2 * c, 1),
                    ConvUnit(q + ".ffn.1", 2 * c, c, 1, act=False),
                ]
        elif sp.kind == "Detect":
            c2, c3 = detect_mid_channels(sp.c1, sp.nc)
            for lvl, x in enumerate(sp.c1):
                q = f"{p}.cv2.{lvl}"
                out += [ConvUnit(q + ".0", x, c2, 3), ConvUnit(q + ".1", c2, c2, 3),
                        ConvUnit(q + ".2", c2, 64, 1, act=False, bn=False)]
            for lvl, x in enumerate(sp.c1):
                q = f"{p}.cv3.{lvl}"
                out += [ConvUnit(q + ".0.0", x, x, 3, g=x), ConvUnit(q + ".0.1", x, c3, 1),
                        ConvUnit(q + ".1.0", c3, c3, 3, g=c3), ConvUnit(q + ".1.1", c3, c3, 1),
                        ConvUnit(q + ".2", c3, sp.nc, 1, act=False, bn=False)]
    return out


def count_parameters(specs: Sequence[LayerSpec]) -> int:
    """Parameter count as ``sum(p.numel())`` of the reference module (yolo11.yaml:8-12 known answers)."""
    total = 0
    for u in conv_units(specs):
        total += u.c2 * (u.c1 // u.g) * u.k * u.k
        total += 2 * u.c2 if u.bn else u.c2
    return total + 16  # DFL arange conv (block.py:60-64)
