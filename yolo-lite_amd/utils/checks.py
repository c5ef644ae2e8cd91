"""Argument checks of the predict path (reference ``yololite/utils/checks.py``)."""
from __future__ import annotations

import ast
import math

import torch

from . import LOGGER


def check_imgsz(imgsz, stride=32, min_dim=1, max_dim=2, floor=0):
    """Round image size(s) up to a multiple of ``stride`` (reference checks.py:80-128)."""
    stride = int(stride.max() if isinstance(stride, torch.Tensor) else stride)
    if isinstance(imgsz, int):
        imgsz = [imgsz]
    elif isinstance(imgsz, (list, tuple)):
        imgsz = list(imgsz)
    elif isinstance(imgsz, str):
        imgsz = [int(imgsz)] if imgsz.isnumeric() else list(ast.literal_eval(imgsz))
    else:
        raise TypeError(f"'imgsz={imgsz}' is of invalid type {type(imgsz).__name__}. Valid imgsz types are int or list")
    if len(imgsz) > max_dim:
        if max_dim != 1:
            raise ValueError(f"imgsz={imgsz} is not a valid image size.")
        imgsz = [max(imgsz)]
    sz = [max(math.ceil(x / stride) * stride, floor) for x in imgsz]
    if sz != imgsz:
        LOGGER.warning(f"WARNING imgsz={imgsz} must be multiple of max stride {stride}, updating to {sz}")
    return [sz[0], sz[0]] if min_dim == 2 and len(sz) == 1 else sz[0] if min_dim == 1 and len(sz) == 1 else sz
