"""-m gpu: BASELINE.json config 1 end to end -- an image FILE at the geometry of the reference's boats.jpg
(1080 x 1920; engine/predictor.py:67-127, data/augment.py:636-681): PIL-encoded JPEG on disk ->
YOLOLite.predict(path, imgsz=640) -> LetterBox(auto) to 384 x 640, batch 1 -> network -> NMS -> scale_boxes back to
1080 x 1920 -> Results.  Checked against the oracle run on the same decoded pixels letterboxed by the independent float
definition of INTER_LINEAR (oracle/letterbox_oracle.py), with the post-NMS gate of oracle/parity_gate.py."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import letterbox_oracle as lbo
from oracle import yolo11_oracle as oracle
from oracle.parity_gate import parity_gate

pytestmark = pytest.mark.gpu


def _scene(h=1080, w=1920, seed=11):
    """A smooth synthetic scene with structure at several scales (JPEG-friendly, unlike white noise)."""
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for _ in range(40):
        cx, cy, r = g.uniform(0, w), g.uniform(0, h), g.uniform(20, 300)
        col = g.uniform(0, 255, 3)
        m = np.exp(-(((xx - cx) / r) ** 2 + ((yy - cy) / r) ** 2) * 2.0)
        img += m[..., None] * col
    img = img / img.max() * 255.0
    return np.clip(img + g.normal(0, 6, img.shape), 0, 255).astype(np.uint8)


def test_predict_jpeg_file_1080x1920(gpu_device, golden, tmp_path):
    from yololite_amd import YOLOLite
    from yololite_amd.nn.synth import calibrate_cls_bias, synth_state_dict

    path = tmp_path / "boats_like.jpg"
    Image.fromarray(_scene()).save(path, quality=92)
    bgr = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])  # what the loader decodes
    assert bgr.shape == (1080, 1920, 3)
    x8 = lbo.letterbox_batch_u8([bgr], (640, 640), auto=True)  # predictor.py:106-107: auto = same shapes and pt
    assert x8.shape == (1, 3, 384, 640)  # SURVEY a3 [probe]: resize (640, 360), pads top / bottom 12
    assert (x8[0, :, :12] == 114).all() and (x8[0, :, -12:] == 114).all() and not (x8[0, :, 12] == 114).all()
    x = torch.from_numpy(x8).float() / 255.0  # predictor.py:82-84

    # synthetic weights whose class bias is calibrated ON THE ORACLE so that ~1.5 % of the anchors clear conf 0.25
    sd0 = synth_state_dict("n", 0, cls_bias=[0.0, 0.0, 0.0], cls_gain=8.0)
    with torch.inference_mode():
        _, raw0 = oracle.forward(sd0, x, "n")
    bias = calibrate_cls_bias([r[:, 64:].amax(1) for r in raw0], 0.015)
    sd = synth_state_dict("n", 0, cls_bias=bias, cls_gain=8.0)

    model = YOLOLite("yolo11n.yaml")
    model.model.load_state_dict(sd)
    res = model.predict(str(path), imgsz=640, conf=0.25, iou=0.7, device=str(gpu_device), save=False, verbose=False)
    assert len(res) == 1 and res[0].orig_shape == (1080, 1920) and res[0].path == str(path)
    shapes = {k[:3] for k in model.predictor.model._engines}
    assert shapes == {(1, 384, 640)}, shapes  # B = 1 at 384 x 640, not 640 x 640
    got = res[0].boxes.data.cpu()
    assert got.shape[1] == 6 and len(got) > 0
    assert float(got[:, [0, 2]].max()) <= 1920 and float(got[:, [1, 3]].max()) <= 1080 and float(got[:, :4].min()) >= 0

    def oracle_dets(storage):
        with torch.inference_mode():
            y, _ = oracle.forward(sd, x, "n", storage=storage)
        d = oracle.non_max_suppression(y, 0.25, 0.7)[0]
        d[:, :4] = oracle.scale_boxes((384, 640), d[:, :4], (1080, 1920))  # predictor.py:125
        return d, y

    # (a) against the bf16-storage oracle: the kernels' own deviation (+ <= 1 LSB of letterbox fixed point)
    d16, _ = oracle_dets("bf16")
    st = parity_gate([d16], [got], 0.25, 0.7, band=0.03, verbose=True)
    print("config 1 vs bf16-storage oracle:", st)
    assert st["ok"], st
    # (b) against the fp32 reference restatement: the stated bf16 format tolerance
    d32, y32 = oracle_dets("fp32")
    st32 = parity_gate([d32], [got], 0.25, 0.7, band=0.08, ref_y=y32, verbose=True)
    print("config 1 vs fp32 oracle:", st32)
    assert st32["ok"], st32
