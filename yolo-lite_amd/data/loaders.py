"""Inference sources: tensors, PIL / numpy images and image files (reference ``yololite/data/loaders.py``).

Video files, streams and screenshots are I/O features outside the accelerated path (SURVEY.md
section 8, out of scope) and raise ``NotImplementedError``.
"""
from __future__ import annotations

import glob
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from ..utils import LOGGER

IMG_FORMATS = {"bmp", "dng", "jpeg", "jpg", "mpo", "png", "tif", "tiff", "webp", "pfm"}
VID_FORMATS = {"asf", "avi", "gif", "m4v", "mkv", "mov", "mp4", "mpeg", "mpg", "ts", "wmv", "webm"}


@dataclass
class SourceTypes:
    stream: bool = False
    screenshot: bool = False
    from_img: bool = False
    tensor: bool = False


def _pil_to_bgr(im: Image.Image) -> np.ndarray:
    if im.mode != "RGB":
        im = im.convert("RGB")
    return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])


class _SingleBatch:
    """A source that is ONE batch: iterating it yields a single ``(paths, images, info)`` triple.

    The two in-memory sources of the reference (``LoadTensor`` loaders.py:480-548, ``LoadPilAndNumpy`` loaders.py:415-477)
    are both this: everything is decided in the constructor, the iterator only hands the batch over once.  Subclasses
    fill ``im0`` (the batch), ``paths`` and ``bs`` through ``_accept``."""

    mode = "image"
    single_batch = True  # nothing for the predictor to pipeline

    def _accept(self, batch, paths):
        self.im0, self.paths, self.bs = batch, list(paths), len(paths)

    def __iter__(self):
        yield self.paths, self.im0, [""] * self.bs

    def __len__(self):
        return self.bs


def _names(items):
    # a PIL image opened from a file carries its file name; everything else is called image<i>.jpg (loaders.py:433,509)
    return [getattr(it, "filename", "") or f"image{i}.jpg" for i, it in enumerate(items)]


class LoadTensor(_SingleBatch):
    """BCHW tensor, H and W multiples of 32; float tensors above 1 are taken to be 0..255 (reference loaders.py:511-532)."""

    STRIDE = 32

    def __init__(self, im0) -> None:
        t = self.checked(im0)
        # (a tensor's images carry no file names: image<i>.jpg each -- without walking the tensor, which unbinds it into B views)
        self._accept(t, [f"image{i}.jpg" for i in range(t.shape[0])])

    @classmethod
    def checked(cls, t: torch.Tensor) -> torch.Tensor:
        complaint = (f"WARNING torch.Tensor inputs should be BCHW i.e. shape(1, 3, 640, 640) divisible by stride {cls.STRIDE}. "
                     f"Input shape{tuple(t.shape)} is incompatible.")
        if t.dim() == 3:  # a single CHW image is tolerated with a warning
            LOGGER.warning(complaint)
            t = t[None]
        if t.dim() != 4 or any(int(n) % cls.STRIDE for n in t.shape[2:]):
            raise ValueError(complaint)
        if t.dtype.is_floating_point:
            top = float(t.max())
            if top > 1.0 + torch.finfo(t.dtype).eps:
                LOGGER.warning(f"WARNING torch.Tensor inputs should be normalized 0.0-1.0 but max value is {top}. Dividing input by 255.")
                t = t.float() / 255.0
        return t


class LoadTensorBatches:
    """A SEQUENCE of BCHW tensors, each one batch (same checks as ``LoadTensor`` per element).  Not in the reference,
    whose tensor source is always a single batch (loaders.py:539-544): it exists so that ``predict(stream=True)`` has
    something to pipeline when the images already live on the device -- pass an instance as ``source`` (like every
    constructed loader, build.py:150-152)."""

    mode = "image"
    source_type = SourceTypes(tensor=True)
    resident = True  # the tensors exist before iteration starts: the predictor orders them in front of its batches ONCE

    def __init__(self, batches) -> None:
        self.batches = [LoadTensor.checked(t) for t in batches]
        if not self.batches:
            raise ValueError("LoadTensorBatches needs at least one BCHW tensor")
        self.bs = int(self.batches[0].shape[0])

    def __iter__(self):
        for k, t in enumerate(self.batches):
            n = int(t.shape[0])
            yield [f"image{k * self.bs + i}.jpg" for i in range(n)], t, [""] * n

    def __len__(self):
        return len(self.batches)


class LoadPilAndNumpy(_SingleBatch):
    """One PIL image / BGR HWC array, or a list of them, as one batch of BGR arrays (reference loaders.py:426-461)."""

    def __init__(self, im0):
        items = im0 if isinstance(im0, list) else [im0]
        self._accept([self.as_bgr(it) for it in items], _names(items))

    @staticmethod
    def as_bgr(it) -> np.ndarray:
        if isinstance(it, Image.Image):
            return _pil_to_bgr(it)
        if isinstance(it, np.ndarray):
            return it
        raise AssertionError(f"Expected PIL/np.ndarray image type, but got {type(it)}")


def _expand_source(spec) -> list:
    """One source spec -- a file, a directory, a glob pattern, a .txt list of those, or a list / tuple of them -- as a sorted list
    of absolute file names (what the reference's loader walks, loaders.py:268-292)."""
    if isinstance(spec, (list, tuple)):
        return [f for item in sorted(spec) for f in _expand_source(item)]
    here = Path(spec)
    if isinstance(spec, str) and here.suffix == ".txt":
        return _expand_source(here.read_text().splitlines())  # (a list: sorted, as the reference does)
    full = str(here.absolute())
    if "*" in full:
        return sorted(glob.glob(full, recursive=True))
    if here.is_dir():
        return sorted(str(q) for q in here.absolute().glob("*.*"))
    if here.is_file():
        return [full]
    raise FileNotFoundError(f"{spec} does not exist")


class LoadImagesAndVideos:
    """Image files / directories / globs in batches of ``batch`` (reference loaders.py:248-412; video files are outside the
    accelerated predict path and raise).  Iteration is a generator over slices of the file list: every batch is
    ``(paths, BGR arrays, log prefixes)``."""

    mode = "image"

    def __init__(self, path, batch=1, vid_stride=1):
        by_kind = {"image": [], "video": []}
        for f in _expand_source(path):
            ext = f.rsplit(".", 1)[-1].lower()
            if ext in IMG_FORMATS:
                by_kind["image"].append(f)
            elif ext in VID_FORMATS:
                by_kind["video"].append(f)
        if by_kind["video"]:
            raise NotImplementedError("video sources are outside the accelerated predict path")
        if not by_kind["image"]:
            raise FileNotFoundError(f"No images found in {path}")
        self.files, self.bs = by_kind["image"], int(batch)
        self.nf = len(self.files)

    @staticmethod
    def _read(path: str) -> np.ndarray:
        try:
            return _pil_to_bgr(Image.open(path))
        except Exception as e:  # same contract as the reference: an unreadable image is a FileNotFoundError
            raise FileNotFoundError(f"Image Read Error {path}") from e

    def __iter__(self):
        for first in range(0, self.nf, self.bs):
            names = self.files[first: first + self.bs]
            yield names, [self._read(f) for f in names], [f"image {first + k + 1}/{self.nf} {f}: " for k, f in enumerate(names)]

    def __len__(self):
        return -(-self.nf // self.bs)


def autocast_list(source):
    """A list of paths / PIL images / arrays as a list of PIL images / arrays (reference loaders.py:551-564)."""
    def one(item):
        if isinstance(item, (Image.Image, np.ndarray)):
            return item
        if isinstance(item, (str, Path)):
            if str(item).startswith("http"):
                raise NotImplementedError("remote sources need network access")
            return Image.open(item)
        raise TypeError(f"type {type(item).__name__} is not a supported prediction source type.")

    return [one(item) for item in source]


LOADERS = (LoadTensor, LoadTensorBatches, LoadPilAndNumpy, LoadImagesAndVideos)
