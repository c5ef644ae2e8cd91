"""Inference sources: tensors, PIL / numpy images and image files (reference ``yololite/data/loaders.py``).

Video files, streams and screenshots are I/O features outside the accelerated path (SURVEY.md
section 8, out of scope) and raise ``NotImplementedError``.
"""
from __future__ import annotations

import glob
import math
import os
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


class LoadTensor:
    """BCHW float tensor in [0, 1], H and W divisible by 32 (reference loaders.py:480-548)."""

    def __init__(self, im0) -> None:
        self.im0 = self._single_check(im0)
        self.bs = self.im0.shape[0]
        self.mode = "image"
        self.paths = [getattr(im, "filename", f"image{i}.jpg") for i, im in enumerate(im0)]

    @staticmethod
    def _single_check(im, stride=32):
        s = (f"WARNING torch.Tensor inputs should be BCHW i.e. shape(1, 3, 640, 640) "
             f"divisible by stride {stride}. Input shape{tuple(im.shape)} is incompatible.")
        if len(im.shape) != 4:
            if len(im.shape) != 3:
                raise ValueError(s)
            LOGGER.warning(s)
            im = im.unsqueeze(0)
        if im.shape[2] % stride or im.shape[3] % stride:
            raise ValueError(s)
        if im.dtype.is_floating_point and im.max() > 1.0 + torch.finfo(im.dtype).eps:
            LOGGER.warning(f"WARNING torch.Tensor inputs should be normalized 0.0-1.0 but max value is {im.max()}. "
                           f"Dividing input by 255.")
            im = im.float() / 255.0
        return im

    def __iter__(self):
        self.count = 0
        return self

    def __next__(self):
        if self.count == 1:
            raise StopIteration
        self.count += 1
        return self.paths, self.im0, [""] * self.bs

    def __len__(self):
        return self.bs


class LoadPilAndNumpy:
    """PIL images / BGR HWC numpy arrays as one batch (reference loaders.py:415-477)."""

    def __init__(self, im0):
        if not isinstance(im0, list):
            im0 = [im0]
        self.paths = [getattr(im, "filename", "") or f"image{i}.jpg" for i, im in enumerate(im0)]
        self.im0 = [self._single_check(im) for im in im0]
        self.mode = "image"
        self.bs = len(self.im0)

    @staticmethod
    def _single_check(im):
        assert isinstance(im, (Image.Image, np.ndarray)), f"Expected PIL/np.ndarray image type, but got {type(im)}"
        if isinstance(im, Image.Image):
            im = _pil_to_bgr(im)
        return im

    def __len__(self):
        return len(self.im0)

    def __iter__(self):
        self.count = 0
        return self

    def __next__(self):
        if self.count == 1:
            raise StopIteration
        self.count += 1
        return self.paths, self.im0, [""] * self.bs


class LoadImagesAndVideos:
    """Image files / directories / globs in batches (reference loaders.py:248-412, images only)."""

    def __init__(self, path, batch=1, vid_stride=1):
        if isinstance(path, str) and Path(path).suffix == ".txt":
            path = Path(path).read_text().splitlines()
        files = []
        for p in sorted(path) if isinstance(path, (list, tuple)) else [path]:
            a = str(Path(p).absolute())
            if "*" in a:
                files.extend(sorted(glob.glob(a, recursive=True)))
            elif os.path.isdir(a):
                files.extend(sorted(glob.glob(os.path.join(a, "*.*"))))
            elif os.path.isfile(a):
                files.append(a)
            else:
                raise FileNotFoundError(f"{p} does not exist")
        images = [f for f in files if f.split(".")[-1].lower() in IMG_FORMATS]
        videos = [f for f in files if f.split(".")[-1].lower() in VID_FORMATS]
        if videos:
            raise NotImplementedError("video sources are outside the accelerated predict path")
        if not images:
            raise FileNotFoundError(f"No images found in {path}")
        self.files, self.nf, self.bs, self.mode = images, len(images), batch, "image"

    def __iter__(self):
        self.count = 0
        return self

    def __next__(self):
        paths, imgs, info = [], [], []
        while len(imgs) < self.bs:
            if self.count >= self.nf:
                if imgs:
                    return paths, imgs, info
                raise StopIteration
            path = self.files[self.count]
            try:
                im0 = _pil_to_bgr(Image.open(path))
            except Exception as e:  # same contract as the reference: unreadable image
                raise FileNotFoundError(f"Image Read Error {path}") from e
            self.count += 1
            paths.append(path)
            imgs.append(im0)
            info.append(f"image {self.count}/{self.nf} {path}: ")
        return paths, imgs, info

    def __len__(self):
        return math.ceil(self.nf / self.bs)


def autocast_list(source):
    """List of paths / PIL / numpy -> list of PIL / numpy (reference loaders.py:551-564)."""
    files = []
    for im in source:
        if isinstance(im, (str, Path)):
            if str(im).startswith("http"):
                raise NotImplementedError("remote sources need network access")
            files.append(Image.open(im))
        elif isinstance(im, (Image.Image, np.ndarray)):
            files.append(im)
        else:
            raise TypeError(f"type {type(im).__name__} is not a supported prediction source type.")
    return files


LOADERS = (LoadTensor, LoadPilAndNumpy, LoadImagesAndVideos)
