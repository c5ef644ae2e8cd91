"""-m gpu: BASELINE.json config 1 end to end -- an image FILE at the geometry of the reference's boats.jpg
(1080 x 1920; engine/predictor.py:67-127, data/augment.py:636-681): PIL-encoded JPEG on disk ->
YOLOLite.predict(path, imgsz=640) -> LetterBox(auto) to 384 x 640, batch 1 -> network -> NMS -> scale_boxes back to
1080 x 1920 -> Results.  Checked against the oracle run on the same decoded pixels letterboxed by the independent float
definition of INTER_LINEAR (oracle/letterbox_oracle.py), with the post-NMS gate of oracle/parity_gate.py."""
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import letterbox_oracle as lbo
from oracle import yolo11_oracle as oracle
from oracle.parity_gate import format_yardstick_ok, parity_gate

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


BOATS = Path(__file__).resolve().parent / "golden" / "boats.jpg"  # the reference's own example image (a data file, 1080 x 1920)


@pytest.mark.parametrize("source", ["synthetic_scene", "boats.jpg"])
def test_predict_jpeg_file_1080x1920(gpu_device, golden, tmp_path, source):
    from yololite_amd import YOLOLite
    from yololite_amd.nn.synth import calibrate_cls_bias, synth_state_dict

    if source == "boats.jpg":  # BASELINE.json configs[0] names this file (reference main.py / README: predict on boats.jpg)
        path = tmp_path / "boats.jpg"
        path.write_bytes(BOATS.read_bytes())
    else:
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

    # the product's letterboxed input (GPU kernel, OpenCV's fixed point) against the float definition: <= 1 LSB
    xin = model.predictor.preprocess([bgr]).cpu().numpy().astype(np.int32)
    assert xin.shape == (1, 3, 384, 640) and np.abs(xin - x8.astype(np.int32)).max() <= 1

    # pre-NMS tensor (the reference's three-stage path keeps it): fixed format tolerances
    model.predictor.fuse_postprocess = False
    res3 = model.predict(str(path), imgsz=640, conf=0.25, iou=0.7, device=str(gpu_device), save=False, verbose=False)
    assert torch.equal(res3[0].boxes.data.cpu(), got), "fused and staged predict paths disagree"
    yh = model.predictor.last_preds.cpu()
    d32, y32 = oracle_dets("fp32")
    st = torch.tensor([float(s) for s in (8, 16, 32) for _ in range((384 // s) * (640 // s))])
    assert yh.shape == y32.shape == (1, 84, 5040)  # A = 5040 at 384 x 640
    assert float((yh[:, 4:] - y32[:, 4:]).abs().max()) < 0.08
    assert float(((yh[:, :4] - y32[:, :4]).abs() / st).max()) < 0.35

    # post-processing in isolation: the oracle's NMS + scale_boxes applied to the HIP network's own pre-NMS tensor must
    # reproduce the product's Results bit for bit (the network is judged above with tolerances, NMS exactly)
    exact = oracle.non_max_suppression(yh, 0.25, 0.7)[0]
    exact[:, :4] = oracle.scale_boxes((384, 640), exact[:, :4], (1080, 1920))
    assert torch.equal(exact, got), "NMS / scale_boxes of the predict path differ from the oracle on identical inputs"

    # post-NMS end to end against the fp32 restatement, next to what the bf16 format costs on this very image.  One image
    # gives ~30 boxes, many of them clusters of near-identical candidates from neighbouring anchors (a smooth scene), so a
    # pair may join boxes of DIFFERENT anchors and its |dconf| is not bounded by the per-anchor tolerance checked above:
    # classes exact, nothing unexplained and the paired fraction are required, max |dconf| is reported and bounded loosely.
    d16, _ = oracle_dets("bf16")
    st_hip = parity_gate([d32], [got], 0.25, 0.7, band=0.08, ref_y=y32, min_matched_frac=0.0, verbose=True)
    st_fmt = parity_gate([d32], [d16], 0.25, 0.7, band=0.08, ref_y=y32, min_matched_frac=0.0)
    verdict = format_yardstick_ok(st_hip, st_fmt, 0.08)
    print("config 1: HIP", st_hip, "\n          format", st_fmt, "\n         ", verdict)
    ck = verdict["checks"]
    assert ck["class_exact"] and ck["nothing_unexplained"] and ck["matched_frac_vs_format"] and ck["paired_something"], verdict
    assert st_hip["max_dconf"] <= 0.12


def test_predict_profile_true_prints_the_per_layer_table(gpu_device, caplog):
    """``predict(profile=True)``: the reference's per-layer profiler (nn/tasks.py:155-177: time, GFLOPs, params per yaml layer, then
    the total) on the HIP path -- every launch of the hot path timed with HIP events (yl_engine_profile_detect) and summed per
    layer.  The table accounts for the whole network: the layers' GFLOPs add up to the architecture's known 6.48 GFLOP per
    640 x 640 image scaled to this input, the parameters to YOLO11n's 2 624 080 (yolo11.yaml:8), and the results of
    the call are those of a plain predict."""
    import logging

    from yololite_amd import YOLOLite

    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(3)).to(gpu_device)
    m = YOLOLite("yolo11n.yaml")
    plain = m.predict(x, conf=0.25, save=False, verbose=False)
    with caplog.at_level(logging.INFO):
        res = m.predict(x, conf=0.25, save=False, verbose=False, profile=True)
    rows = m.predictor.layer_profile
    assert [r["layer"] for r in rows] == list(range(24)) + ["post"]
    assert rows[23]["module"] == "Detect" and rows[23]["launches"] >= 4 and rows[-1]["launches"] in (1, 4)  # (one launch from batch 16 up, else decode, sort, mask, scan)
    assert all(r["ms"] >= 0 for r in rows) and sum(r["ms"] for r in rows) > 0
    gflops = sum(r["gflops"] for r in rows)
    assert abs(gflops - 2 * 6.4811 * (96 * 128) / (640 * 640)) / gflops < 0.02, gflops  # two images
    assert sum(r["params"] for r in rows) + 16 == 2624080  # (+ the DFL's frozen arange conv, block.py:60-64)
    assert all(torch.equal(a.boxes.data, b.boxes.data) for a, b in zip(plain, res))
