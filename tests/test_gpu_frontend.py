"""-m gpu: the device letterbox (yl_letterbox_u8) against the CPU restatement (data/augment.py), bit for bit,
and the predictor's image-list path through it."""
import numpy as np
import pytest
import torch

from yololite_amd.data.augment import DeviceLetterBox, LetterBox

pytestmark = pytest.mark.gpu


def _cpu_batch(images, new_shape, auto, stride=32):
    lb = LetterBox(new_shape, auto=auto, stride=stride)
    x = np.stack([lb(image=im) for im in images])
    return np.ascontiguousarray(x[..., ::-1].transpose(0, 3, 1, 2))  # predictor.py:77-81


@pytest.mark.parametrize("h,w,new_shape,auto", [(1080, 1920, (640, 640), True), (1080, 1920, (640, 640), False),
                                                 (480, 640, (640, 640), True), (640, 481, (640, 640), False),
                                                 (500, 381, (640, 640), True), (37, 91, (96, 128), False),
                                                 (640, 640, (640, 640), False), (200, 300, (320, 320), True)])
def test_device_letterbox_bit_exact(gpu_device, h, w, new_shape, auto):
    g = np.random.default_rng(h * 7 + w)
    images = [g.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(3)]
    ref = _cpu_batch(images, new_shape, auto)
    got = DeviceLetterBox(new_shape, auto=auto, device=gpu_device)(images).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.uint8
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("h,w,new_shape,auto", [(1080, 1920, (640, 640), True), (480, 640, (640, 640), True),
                                                 (500, 381, (640, 640), True), (37, 91, (96, 128), False)])
def test_device_letterbox_within_one_lsb_of_float_definition(gpu_device, h, w, new_shape, auto):
    """The HIP kernel against the INDEPENDENT float64 definition of INTER_LINEAR (oracle/letterbox_oracle.py), not against
    the product's own integer restatement: at most 1 LSB (OpenCV's 11-bit fixed-point rounding), same geometry and
    channel order."""
    from oracle import letterbox_oracle as lbo

    g = np.random.default_rng(h * 11 + w)
    images = [g.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(2)]
    got = DeviceLetterBox(new_shape, auto=auto, device=gpu_device)(images).cpu().numpy().astype(np.float64)
    ref = np.stack([lbo.letterbox_f64(im, new_shape, auto=auto) for im in images])[..., ::-1].transpose(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1.0


def test_device_letterbox_rejects_mixed_shapes(gpu_device):
    g = np.random.default_rng(0)
    images = [g.integers(0, 256, (100, 200, 3), dtype=np.uint8), g.integers(0, 256, (200, 100, 3), dtype=np.uint8)]
    with pytest.raises(ValueError):
        DeviceLetterBox((640, 640), auto=True, device=gpu_device)(images)


def test_predict_image_list_uses_device_letterbox(gpu_device):
    """YOLOLite.predict on numpy images: same detections as feeding the CPU-letterboxed uint8 batch as a tensor."""
    from yololite_amd import YOLOLite

    g = np.random.default_rng(5)
    images = [g.integers(0, 256, (270, 480, 3), dtype=np.uint8) for _ in range(2)]
    model = YOLOLite("yolo11n.yaml")
    res = model.predict(images, conf=0.25, iou=0.7, imgsz=320, device=str(gpu_device), save=False, verbose=False)
    assert len(res) == 2 and all(r.orig_shape == (270, 480) for r in res)
    x = torch.from_numpy(_cpu_batch(images, (320, 320), True)).to(gpu_device)  # auto: same shapes, pt model
    assert tuple(x.shape[2:]) == (192, 320)
    eng = model.predictor.model.engine_for(2, x.shape[2], x.shape[3])
    sc = []
    from yololite_amd.utils import ops
    for _ in range(2):
        gain, pad = ops.scale_boxes_params((192, 320), (270, 480))
        sc.append([pad[0], pad[1], gain, 480, 270])
    out, cnt = eng.detect(x, 0.25, 0.7, max_det=300, scale=torch.tensor(sc, dtype=torch.float32, device=gpu_device), graph=False)
    for i, r in enumerate(res):
        assert torch.equal(r.boxes.data, out[i, : int(cnt[i])])


def test_predict_saves_annotated_images_and_crops(gpu_device, tmp_path):
    """save=True / save_crop=True / save_txt=True on an image source write the reference's files (PIL-drawn)."""
    from yololite_amd import YOLOLite

    g = np.random.default_rng(9)
    images = [g.integers(0, 256, (200, 320, 3), dtype=np.uint8) for _ in range(2)]
    model = YOLOLite("yolo11n.yaml")
    res = model.predict(images, conf=0.05, imgsz=320, device=str(gpu_device), save=True, save_crop=True, save_txt=True,
                        project=str(tmp_path), name="run", verbose=False)
    d = tmp_path / "run"
    assert sorted(p.name for p in d.glob("*.jpg")) == ["image0.jpg", "image1.jpg"]
    n = sum(len(r) for r in res)
    if n:
        assert len(list((d / "crops").rglob("*.jpg"))) == n
        assert sum(len(f.read_text().strip().splitlines()) for f in (d / "labels").glob("*.txt")) == n
