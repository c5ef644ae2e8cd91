"""Source-type dispatch (reference ``yololite/data/build.py:119-176``)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from PIL import Image

from .loaders import (IMG_FORMATS, LOADERS, VID_FORMATS, LoadImagesAndVideos, LoadPilAndNumpy, LoadTensor, SourceTypes,
                      autocast_list)


def check_source(source):
    webcam = screenshot = from_img = in_memory = tensor = False
    if isinstance(source, (str, int, Path)):
        source = str(source)
        is_file = Path(source).suffix[1:] in (IMG_FORMATS | VID_FORMATS)
        is_url = source.lower().startswith(("https://", "http://", "rtsp://", "rtmp://", "tcp://"))
        webcam = source.isnumeric() or source.endswith(".streams") or (is_url and not is_file)
        screenshot = source.lower() == "screen"
    elif isinstance(source, LOADERS):
        in_memory = True
    elif isinstance(source, (list, tuple)):
        source = autocast_list(source)
        from_img = True
    elif isinstance(source, (Image.Image, np.ndarray)):
        from_img = True
    elif isinstance(source, torch.Tensor):
        tensor = True
    else:
        raise TypeError("Unsupported image type. Supported: str/Path, PIL.Image, np.ndarray, list/tuple, torch.Tensor")
    return source, webcam, screenshot, from_img, in_memory, tensor


def load_inference_source(source=None, batch=1, vid_stride=1, buffer=False):
    source, stream, screenshot, from_img, in_memory, tensor = check_source(source)
    source_type = source.source_type if in_memory else SourceTypes(stream, screenshot, from_img, tensor)
    if tensor:
        dataset = LoadTensor(source)
    elif in_memory:
        dataset = source
    elif stream or screenshot:
        raise NotImplementedError("stream / screenshot sources are outside the accelerated predict path")
    elif from_img:
        dataset = LoadPilAndNumpy(source)
    else:
        dataset = LoadImagesAndVideos(source, batch=batch, vid_stride=vid_stride)
    setattr(dataset, "source_type", source_type)
    return dataset
