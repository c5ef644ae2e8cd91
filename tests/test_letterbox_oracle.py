"""CPU: the product's fixed-point LetterBox pixels (data/augment.py::resize_linear_u8, the arithmetic the HIP kernel
yl_letterbox_u8 repeats bit for bit) against the independent float64 definition of INTER_LINEAR in
oracle/letterbox_oracle.py: at most 1 LSB apart (the 11-bit coefficient rounding), geometry identical."""
import numpy as np
import pytest

from oracle import letterbox_oracle as lbo
from yololite_amd.data.augment import LetterBox, resize_linear_u8

GEOMETRIES = [(1080, 1920, (640, 640), True), (1080, 1920, (640, 640), False), (480, 640, (640, 640), True),
              (640, 481, (640, 640), False), (500, 381, (640, 640), True), (37, 91, (96, 128), False),
              (640, 640, (640, 640), False), (200, 300, (320, 320), True)]


@pytest.mark.parametrize("h,w,new_shape,auto", GEOMETRIES)
def test_fixed_point_letterbox_within_one_lsb_of_float_definition(h, w, new_shape, auto):
    g = np.random.default_rng(h * 7 + w)
    img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = lbo.letterbox_f64(img, new_shape, auto=auto)
    got = LetterBox(new_shape, auto=auto, stride=32)(image=img)
    assert got.shape == ref.shape and got.dtype == np.uint8
    err = np.abs(got.astype(np.float64) - ref)
    assert err.max() <= 1.0, f"max |fixed-point - float| = {err.max():.3f} LSB"
    # OpenCV's vertical pass floors twice ((b * s) >> 16 per row, in quarter-LSB units): a known bias of about -1/4 LSB
    assert abs((got.astype(np.float64) - ref).mean()) < 0.3


def test_smooth_image_and_upscale():
    yy, xx = np.mgrid[0:90, 0:160]
    img = np.stack([(xx * 255 // 159), (yy * 255 // 89), ((xx + yy) % 256)], -1).astype(np.uint8)
    for nw, nh in ((640, 360), (320, 180), (161, 91), (160, 90)):
        ref = lbo.resize_bilinear_f64(img, nw, nh) if (nw, nh) != (160, 90) else img.astype(np.float64)
        got = resize_linear_u8(img, nw, nh) if (nw, nh) != (160, 90) else img
        assert np.abs(got.astype(np.float64) - ref).max() <= 1.0


def test_batch_layout_matches_predictor_preprocess():
    g = np.random.default_rng(3)
    imgs = [g.integers(0, 256, (108, 192, 3), dtype=np.uint8) for _ in range(2)]
    x = lbo.letterbox_batch_u8(imgs, (64, 64), auto=True)
    assert x.shape == (2, 3, 64, 64) and x.dtype == np.uint8  # 108x192 -> 36x64 + pads 14/14 (stride-modulo: 64 - 36 = 28)
    assert (x[:, :, 0] == 114).all() and (x[:, :, -1] == 114).all()
    # RGB order: channel 0 of the batch is channel 2 (R) of the BGR source
    one = np.zeros((32, 64, 3), np.uint8)
    one[..., 2] = 200
    y = lbo.letterbox_batch_u8([one], (64, 64), auto=False)
    assert y[0, 0, 32, 32] == 200 and y[0, 2, 32, 32] == 0
