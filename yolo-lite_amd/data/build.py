"""Which loader serves a predict ``source`` (the behaviour of reference ``yololite/data/build.py:119-176``).

The decision is a two-step table lookup instead of the reference's flag tuple: ``classify_source`` names the KIND
of a source, ``_KINDS`` maps each kind to the ``SourceTypes`` flags the predictor reads and to a loader factory.
Stream-like kinds (webcam index, ``.streams`` list, URL, ``"screen"``) are outside the accelerated path and raise.
"""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Dict, NamedTuple, Optional

import numpy as np
import torch
from PIL import Image

from .loaders import (IMG_FORMATS, LOADERS, VID_FORMATS, LoadImagesAndVideos, LoadPilAndNumpy, LoadTensor, SourceTypes,
                      autocast_list)

_URL_PREFIXES = ("https://", "http://", "rtsp://", "rtmp://", "tcp://")


class _Kind(NamedTuple):
    flags: SourceTypes
    make: Optional[Callable]  # (source, batch, vid_stride) -> dataset; None: unsupported here


def _unsupported(what: str):
    def make(source, batch, vid_stride):
        raise NotImplementedError(f"{what} sources are outside the accelerated predict path")

    return make


_KINDS: Dict[str, _Kind] = {
    "tensor": _Kind(SourceTypes(tensor=True), lambda s, b, v: LoadTensor(s)),
    "images": _Kind(SourceTypes(from_img=True), lambda s, b, v: LoadPilAndNumpy(s)),
    "files": _Kind(SourceTypes(), lambda s, b, v: LoadImagesAndVideos(s, batch=b, vid_stride=v)),
    "stream": _Kind(SourceTypes(stream=True), _unsupported("stream")),
    "screen": _Kind(SourceTypes(screenshot=True), _unsupported("screenshot")),
    "loader": _Kind(SourceTypes(), lambda s, b, v: s),  # an already constructed loader: flags come from the object
}


def _path_kind(text: str) -> str:
    """A string / int source: camera index, stream list, URL that is not a media file, the screen, or files."""
    low = text.lower()
    if low == "screen":
        return "screen"
    media_file = Path(text).suffix[1:] in (IMG_FORMATS | VID_FORMATS)
    remote = low.startswith(_URL_PREFIXES)
    if text.isnumeric() or text.endswith(".streams") or (remote and not media_file):
        return "stream"
    return "files"


def classify_source(source):
    """-> (normalised source, kind).  Raises TypeError for anything predict() cannot take (reference build.py:138)."""
    if isinstance(source, torch.Tensor):
        return source, "tensor"
    if isinstance(source, LOADERS):
        return source, "loader"
    if isinstance(source, (Image.Image, np.ndarray)):
        return source, "images"
    if isinstance(source, (list, tuple)):
        return autocast_list(source), "images"
    if isinstance(source, (str, int, Path)):
        text = str(source)
        return text, _path_kind(text)
    raise TypeError("Unsupported image type. Supported: str/Path, PIL.Image, np.ndarray, list/tuple, torch.Tensor")


def check_source(source):
    """The reference's 6-tuple ``(source, webcam, screenshot, from_img, in_memory, tensor)`` for callers that want it."""
    source, kind = classify_source(source)
    f = _KINDS[kind].flags
    return source, f.stream, f.screenshot, f.from_img, kind == "loader", f.tensor


def load_inference_source(source=None, batch=1, vid_stride=1, buffer=False):
    """Dataset for ``source`` with ``.source_type`` attached (reference build.py:143-176)."""
    source, kind = classify_source(source)
    entry = _KINDS[kind]
    dataset = entry.make(source, batch, vid_stride)
    dataset.source_type = source.source_type if kind == "loader" else entry.flags
    return dataset
