"""Result sinks drawn with PIL (reference ``yololite/utils/plotting.py``: ``Colors`` :18-120, ``Annotator``'s
PIL branch, ``save_one_box`` :750-794).  Host-side conveniences outside the accelerated path: the reference
draws with cv2 by default, which is absent here, so only its ``pil=True`` behaviour is mirrored (default PIL
bitmap font instead of Arial.ttf, which would need a download)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import ops

_HEX = ("042AFF", "0BDBEB", "F3F3F3", "00DFB7", "111F68", "FF6FDD", "FF444F", "CCED00", "00F344", "BD00FF",
        "00B4FF", "DD00BA", "00FFFF", "26C000", "01FFB3", "7D24FF", "7B0068", "FF1B6C", "FC6D2F", "A2FF0B")


class Colors:
    """The 20-colour palette of the reference (plotting.py:32-53), indexed modulo its length."""

    def __init__(self):
        self.palette = [tuple(int(h[i: i + 2], 16) for i in (0, 2, 4)) for h in _HEX]
        self.n = len(self.palette)

    def __call__(self, i, bgr=False):
        c = self.palette[int(i) % self.n]
        return (c[2], c[1], c[0]) if bgr else c


colors = Colors()


def increment_path(path: Path) -> Path:
    """im.jpg -> im2.jpg, im3.jpg ... if the file exists (reference utils/files.py increment_path)."""
    path = Path(path)
    if not path.exists():
        return path
    for n in range(2, 9999):
        p = path.with_name(f"{path.stem}{n}{path.suffix}")
        if not p.exists():
            return p
    return path


class Annotator:
    """Boxes + labels on an RGB PIL image (the reference Annotator with pil=True)."""

    def __init__(self, im, line_width=None, font_size=None):
        from PIL import Image, ImageDraw, ImageFont

        self.im = im if isinstance(im, Image.Image) else Image.fromarray(np.ascontiguousarray(im))
        self.draw = ImageDraw.Draw(self.im)
        self.lw = line_width or max(round(sum(self.im.size) / 2 * 0.003), 2)
        size = font_size or max(round(sum(self.im.size) / 2 * 0.035), 12)
        try:
            self.font = ImageFont.load_default(size)
        except TypeError:  # older Pillow: fixed-size bitmap font
            self.font = ImageFont.load_default()

    def box_label(self, box, label="", color=(128, 128, 128), txt_color=(255, 255, 255)):
        box = [float(v) for v in (box.tolist() if hasattr(box, "tolist") else box)]
        self.draw.rectangle(box, width=self.lw, outline=color)
        if label:
            l, t, r, b = self.font.getbbox(label)
            w, h = r - l, b
            outside = box[1] >= h  # label fits above the box
            x0 = box[0]
            if x0 > self.im.size[0] - w:
                x0 = self.im.size[0] - w
            y0 = box[1] - h if outside else box[1]
            self.draw.rectangle((x0, y0, x0 + w + 1, y0 + h + 1), fill=color)
            self.draw.text((x0, y0), label, fill=txt_color, font=self.font)

    def result(self):
        return np.asarray(self.im)


def save_one_box(xyxy, im, file=Path("im.jpg"), gain=1.02, pad=10, square=False, BGR=False, save=True):
    """Crop ``im`` (HWC) around a box grown by ``gain`` and ``pad`` and save it (reference plotting.py:750-794)."""
    from PIL import Image

    if not isinstance(xyxy, torch.Tensor):
        xyxy = torch.as_tensor(np.asarray(xyxy, dtype=np.float32))
    b = ops.xyxy2xywh(xyxy.reshape(-1, 4).float().cpu())
    if square:
        b[:, 2:] = b[:, 2:].max(1)[0].unsqueeze(1)
    b[:, 2:] = b[:, 2:] * gain + pad
    q = ops.xywh2xyxy(b).long()
    q = ops.clip_boxes(q, im.shape)
    crop = im[int(q[0, 1]): int(q[0, 3]), int(q[0, 0]): int(q[0, 2]), :: (1 if BGR else -1)]
    if save:
        file = Path(file)
        file.parent.mkdir(parents=True, exist_ok=True)
        f = str(increment_path(file).with_suffix(".jpg"))
        Image.fromarray(np.ascontiguousarray(crop[..., ::-1])).save(f, quality=95, subsampling=0)
    return crop
