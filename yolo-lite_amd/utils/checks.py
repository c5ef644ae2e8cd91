"""Argument checks of the predict path (behaviour of the reference's ``yololite/utils/checks.py:80-128``)."""
from __future__ import annotations

import ast
from typing import List, Sequence, Union

import torch

from . import LOGGER

Size = Union[int, str, Sequence[int]]


def _size_list(imgsz: Size) -> List[int]:
    """The sizes a caller may pass -- 640, (384, 640), '640', '[384, 640]' -- as a list of ints."""
    if isinstance(imgsz, str):
        text = imgsz.strip()
        if text.isnumeric():
            return [int(text)]
        parsed = ast.literal_eval(text)  # a literal only: the reference evaluates arbitrary expressions here
        return [int(v) for v in parsed] if isinstance(parsed, (list, tuple)) else [int(parsed)]
    if isinstance(imgsz, bool) or not isinstance(imgsz, (int, list, tuple)):
        raise TypeError(f"'imgsz={imgsz}' is of invalid type {type(imgsz).__name__}. Valid imgsz types are int or list")
    return [imgsz] if isinstance(imgsz, int) else list(imgsz)


def _ceil_to(value: float, multiple: int) -> int:
    q, r = divmod(value, multiple)
    return int(q + (1 if r else 0)) * multiple


def check_imgsz(imgsz: Size, stride=32, min_dim: int = 1, max_dim: int = 2, floor: int = 0):
    """Image size(s) rounded UP to a multiple of the model's largest stride and clamped below by ``floor``.

    Returns an int when one size was given and ``min_dim == 1``, a two-element list when ``min_dim == 2`` (a single
    size is duplicated), the rounded list otherwise.  More sizes than ``max_dim``: ``ValueError``, except for
    ``max_dim == 1`` where the largest one is used.  A changed size is reported with a warning, as the reference does.
    """
    step = int(stride.max()) if isinstance(stride, torch.Tensor) else int(stride)
    asked = _size_list(imgsz)
    if len(asked) > max_dim:
        if max_dim != 1:
            raise ValueError(f"imgsz={asked} is not a valid image size.")
        asked = [max(asked)]
    rounded = [max(_ceil_to(v, step), floor) for v in asked]
    if rounded != asked:
        LOGGER.warning(f"WARNING imgsz={asked} must be multiple of max stride {step}, updating to {rounded}")
    if len(rounded) > 1:
        return rounded
    if min_dim == 2:
        return [rounded[0], rounded[0]]
    return rounded[0] if min_dim == 1 else rounded
