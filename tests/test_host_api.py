"""CPU: host-side mirror of the reference API (cfg, sources, letterbox, results, .pt loading)."""
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from yololite_amd import YOLO, YOLOLite
from yololite_amd.cfg import DEFAULT_CFG_DICT, get_cfg
from yololite_amd.data.augment import LetterBox, resize_linear_u8
from yololite_amd.data.build import check_source, load_inference_source
from yololite_amd.data.loaders import LoadImagesAndVideos, LoadPilAndNumpy, LoadTensor
from yololite_amd.engine.results import Boxes, Results
from yololite_amd.nn.tasks import DetectionModel, attempt_load_one_weight
from yololite_amd.utils import ops
from yololite_amd.utils.checks import check_imgsz
from yololite_amd.utils.torch_utils import select_device


def test_cfg_errors_match_reference():
    assert get_cfg(overrides={"conf": 0.3, "iou": 0.5}).conf == 0.3
    with pytest.raises(SyntaxError):          # cfg/__init__.py:239-249
        get_cfg(overrides={"cnof": 0.3})
    with pytest.raises(TypeError):
        get_cfg(overrides={"max_det": 3.5})   # int-only key
    with pytest.raises(ValueError):
        get_cfg(overrides={"iou": 1.5})       # fraction key
    with pytest.raises(TypeError):
        get_cfg(overrides={"half": 1})        # bool-only key
    assert DEFAULT_CFG_DICT["iou"] == 0.7 and DEFAULT_CFG_DICT["max_det"] == 300 and DEFAULT_CFG_DICT["conf"] is None


def test_check_imgsz_and_device():
    assert check_imgsz(640, 32, min_dim=2) == [640, 640]
    assert check_imgsz([600, 630], 32) == [608, 640]
    assert check_imgsz("[320,640]", 32) == [320, 640]
    assert check_imgsz("640", 32) == 640 and check_imgsz(641, 32) == 672
    assert check_imgsz(100, torch.tensor([8.0, 16.0, 32.0]), floor=160) == 160
    assert check_imgsz([600, 630], 32, max_dim=1) == 640   # the larger one wins when only one size is allowed
    assert check_imgsz([64], 32, min_dim=0) == [64]
    with pytest.raises(ValueError):
        check_imgsz([320, 640, 960], 32)
    with pytest.raises(TypeError):
        check_imgsz(None, 32)
    assert select_device("cpu", verbose=False).type == "cpu"
    if not torch.cuda.is_available():
        with pytest.raises(ValueError):
            select_device("0", verbose=False)


def test_sources():
    with pytest.raises(TypeError):
        check_source(3.5)
    with pytest.raises(ValueError):
        LoadTensor(torch.rand(1, 3, 100, 100))      # not divisible by 32 (loaders.py:518-524)
    with pytest.raises(ValueError):
        LoadTensor(torch.rand(3, 3))
    ds = load_inference_source(torch.rand(2, 3, 64, 96))
    paths, im, _ = next(iter(ds))
    assert ds.source_type.tensor and paths == ["image0.jpg", "image1.jpg"] and im.shape == (2, 3, 64, 96)
    pil = Image.fromarray(np.arange(48, dtype=np.uint8).reshape(4, 4, 3))
    ds = load_inference_source([pil, np.zeros((5, 6, 3), np.uint8)])
    _, ims, _ = next(iter(ds))
    assert isinstance(ds, LoadPilAndNumpy) and ims[0].shape == (4, 4, 3) and ims[0][0, 0, 0] == 2  # RGB -> BGR
    with pytest.raises(FileNotFoundError):
        LoadImagesAndVideos("/nonexistent/x.jpg")


def test_letterbox_geometry_and_pixels(golden):
    for h0, w0, auto, nw, nh, top, bottom, left, right, oh, ow, val in golden["letterbox_geo"].tolist():
        lb = LetterBox((640, 640), auto=bool(auto), stride=32)
        new_unpad, pads, _ = lb.geometry((h0, w0))
        assert new_unpad == (nw, nh) and pads == (top, bottom, left, right)
    img = np.random.default_rng(0).integers(0, 256, (108, 192, 3), dtype=np.uint8)
    out = LetterBox((64, 64), auto=True, stride=32)(image=img)
    assert out.shape == (64, 64, 3) and (out[0] == 114).all() and (out[-1] == 114).all()
    # constant images stay constant, identity resize is exact
    assert (resize_linear_u8(np.full((10, 12, 3), 77, np.uint8), 7, 5) == 77).all()
    assert (resize_linear_u8(img, 192, 108) == img).all()
    # a horizontal ramp downscaled 2x lands on the pixel-pair means (half-pixel centres)
    ramp = np.tile(np.arange(0, 64, dtype=np.uint8)[None, :, None], (4, 1, 1))
    r2 = resize_linear_u8(ramp, 32, 4)[0, :, 0].astype(int)
    assert np.abs(r2 - (np.arange(32) * 2 + 0.5)).max() <= 0.5


def test_results_and_boxes():
    data = torch.tensor([[10.0, 20.0, 110.0, 220.0, 0.9, 2.0], [0.0, 0.0, 50.0, 40.0, 0.4, 0.0]])
    r = Results(None, "a.jpg", {0: "person", 1: "x", 2: "boat"}, boxes=data, orig_shape=(400, 200))
    assert len(r) == 2 and r.orig_shape == (400, 200)
    b = r.boxes
    assert torch.equal(b.xyxy, data[:, :4]) and torch.equal(b.conf, data[:, 4]) and torch.equal(b.cls, data[:, 5])
    assert torch.allclose(b.xywh[0], torch.tensor([60.0, 120.0, 100.0, 200.0]))
    assert torch.allclose(b.xyxyn[0], torch.tensor([0.05, 0.05, 0.55, 0.55]))
    assert torch.allclose(b.xywhn[0], torch.tensor([0.3, 0.3, 0.5, 0.5]))
    assert r.verbose() == "1 person, 1 boat, "
    assert isinstance(r.numpy().boxes.data, np.ndarray) and len(r[0]) == 1
    assert r.summary()[0]["name"] == "boat" and "confidence" in r.to_json()
    with pytest.raises(AssertionError):
        Boxes(torch.zeros(2, 5), (4, 4))


def test_scale_boxes_matches_reference(golden):
    for bi, bo, (h1, w1, h0, w0) in zip(golden["scale_boxes_in"], golden["scale_boxes_out"],
                                        golden["scale_boxes_shapes"].tolist()):
        got = ops.scale_boxes((h1, w1), torch.from_numpy(bi.copy()), (h0, w0)).numpy()
        np.testing.assert_array_equal(got, bo)


def test_model_facade():
    m = YOLOLite("yolo11s.yaml")
    assert YOLO is YOLOLite and m.task == "detect" and len(m.names) == 80 and m.names[3] == "3"
    assert m.model.info() == 9458752
    assert m.model.stride.tolist() == [8.0, 16.0, 32.0]
    with pytest.raises(NotImplementedError):
        m.train(data="coco8.yaml")
    with pytest.raises(FileNotFoundError):
        YOLOLite("no_such_weights.pt")
    with pytest.raises(SyntaxError):
        m.predict(torch.rand(1, 3, 64, 64), cnof=1)


def _fake_reference_module_tree(sd, yaml_d, pkg="yololite"):
    """Build nn.Modules under the reference's pickle GLOBAL names (yololite.nn.tasks.DetectionModel,
    yololite.nn.modules.conv.Conv, ...; pkg="ultralytics": the upstream names) so torch.save writes the same
    kind of pickle the reference / upstream does."""
    import torch.nn as nn

    mods = {}
    for name in ("yololite", "yololite.nn", "yololite.nn.tasks", "yololite.nn.modules", "yololite.nn.modules.conv",
                 "yololite.nn.modules.block", "yololite.nn.modules.head"):
        name = name.replace("yololite", pkg)
        mods[name] = types.ModuleType(name)

    def mk(modname, clsname):
        cls = type(clsname, (nn.Module,), {"__module__": modname})
        setattr(mods[modname], clsname, cls)
        return cls

    Generic = mk(f"{pkg}.nn.modules.block", "C3k2")
    Conv = mk(f"{pkg}.nn.modules.conv", "Conv")
    Det = mk(f"{pkg}.nn.tasks", "DetectionModel")
    root = Det()
    # rebuild the hierarchy from the state-dict keys
    for key, val in sd.items():
        parts = key.split(".")
        node = root
        for p_ in parts[:-1]:
            if p_ not in node._modules:
                node.add_module(p_, (Conv if p_ in ("conv", "bn") else Generic)())
            node = node._modules[p_]
        if key.endswith(("running_mean", "running_var", "num_batches_tracked")):
            node.register_buffer(parts[-1], val.clone())
        else:
            node.register_parameter(parts[-1], nn.Parameter(val.clone(), requires_grad=False))
    root.yaml = yaml_d
    root.names = {i: f"cls{i}" for i in range(yaml_d["nc"])}
    root.stride = torch.tensor([8.0, 16.0, 32.0])
    root.anchors_nonpersistent = torch.zeros(3)
    return root, mods


def test_pt_loader_restricted_unpickler(tmp_path):
    from yololite_amd.nn import arch
    from yololite_amd.nn.synth import synth_state_dict

    sd = synth_state_dict("n", 5)
    yaml_d = arch.yolo11_dict("n")
    root, mods = _fake_reference_module_tree(sd, yaml_d)
    sys.modules.update(mods)
    try:
        ckpt = {"epoch": -1, "model": None, "ema": root.half(), "train_args": {"imgsz": 640, "data": "coco8.yaml"},
                "date": "2025-01-01"}
        f = tmp_path / "yolo11n.pt"
        torch.save(ckpt, f)
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    assert "yololite" not in sys.modules
    model, ck = attempt_load_one_weight(str(f))
    assert "yololite" not in sys.modules, "loading must not import (or need) the reference package"
    assert isinstance(model, DetectionModel) and model.names[7] == "cls7" and model.args["imgsz"] == 640
    got = model.state_dict()
    assert set(got) == set(sd)
    for k in sd:
        if sd[k].is_floating_point():
            assert got[k].dtype == torch.float32
            assert torch.equal(got[k], sd[k].half().float()), k   # fp16 checkpoint -> fp32 (tasks.py:503)
    y = YOLOLite(str(f))
    assert y.ckpt_path == str(f) and y.overrides["imgsz"] == 640 and y.names[0] == "cls0"


def test_pt_loader_ultralytics_namespace_and_stripped_checkpoint(tmp_path):
    """an upstream-style checkpoint -- `ultralytics.nn.*` pickle globals, `strip_optimizer` form
    (torch_utils.py:355-399: ema None, optimizer None, fp16 `model`), class names as a list -- loads the same way."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.synth import synth_state_dict

    sd = synth_state_dict("s", 3)
    yaml_d = arch.yolo11_dict("s")
    root, mods = _fake_reference_module_tree(sd, yaml_d, pkg="ultralytics")
    root.names = [f"thing{i}" for i in range(yaml_d["nc"])]
    sys.modules.update(mods)
    try:
        ckpt = {"epoch": -1, "best_fitness": None, "model": root.half(), "ema": None, "updates": None, "optimizer": None,
                "train_args": {"imgsz": 640, "data": "coco.yaml", "model": "yolo11s.pt"}, "version": "8.3.0"}
        f = tmp_path / "yolo11s.pt"
        torch.save(ckpt, f)
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    assert "ultralytics" not in sys.modules
    model, _ = attempt_load_one_weight(str(f))
    assert "ultralytics" not in sys.modules
    assert model.names[79] == "thing79" and model.yaml["scale"] == "s"
    got = model.state_dict()
    assert set(got) == set(sd)
    assert all(torch.equal(got[k], sd[k].half().float()) for k in sd if sd[k].is_floating_point())


def test_pt_loader_blocks_code_execution(tmp_path):
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /tmp/yl_pwned",))

    f = tmp_path / "evil.pt"
    torch.save({"model": Evil(), "yaml": {}}, f)
    import os
    if os.path.exists("/tmp/yl_pwned"):
        os.remove("/tmp/yl_pwned")
    with pytest.raises(Exception):
        attempt_load_one_weight(str(f))
    assert not os.path.exists("/tmp/yl_pwned"), "restricted unpickler executed a pickled callable"


def test_result_sinks_plot_save_crop_txt_json(tmp_path):
    """Results.plot / save / save_crop / save_txt / to_json on a hand-made result."""
    import json

    import numpy as np

    from yololite_amd.engine.results import Results

    img = np.full((120, 160, 3), 30, np.uint8)
    img[40:80, 50:110] = (0, 0, 255)  # a red rectangle (BGR)
    boxes = torch.tensor([[50.0, 40.0, 110.0, 80.0, 0.9, 1.0], [5.0, 5.0, 20.0, 30.0, 0.4, 0.0]])
    r = Results(img, path="a/b/frame.jpg", names={0: "cat", 1: "box"}, boxes=boxes)
    out = r.plot()
    assert out.shape == img.shape and out.dtype == np.uint8 and (out != img).any()
    assert tuple(out[60, 80]) == (0, 0, 255)  # interior untouched, still BGR
    assert tuple(out[40, 80]) != (0, 0, 255)  # the outline was drawn in the class colour
    pil = r.plot(pil=True, conf=False)
    assert pil.size == (160, 120)
    f = r.save(tmp_path / "annotated.png")
    assert (tmp_path / "annotated.jpg").exists() and str(f).endswith("annotated.png")
    r.save_crop(tmp_path / "crops", file_name="x.jpg")
    r.save_crop(tmp_path / "crops", file_name="x.jpg")  # second call increments the name
    assert (tmp_path / "crops" / "box" / "x.jpg").exists() and (tmp_path / "crops" / "box" / "x2.jpg").exists()
    assert (tmp_path / "crops" / "cat" / "x.jpg").exists()
    from PIL import Image
    crop = Image.open(tmp_path / "crops" / "box" / "x.jpg")
    assert crop.size == (round(60 * 1.02 + 10), round(40 * 1.02 + 10))  # gain 1.02, pad 10 (plotting.py:785)
    r.save_txt(tmp_path / "l.txt", save_conf=True)
    lines = (tmp_path / "l.txt").read_text().strip().splitlines()
    assert lines[0].split()[0] == "1" and len(lines[0].split()) == 6 and abs(float(lines[0].split()[1]) - 0.5) < 1e-6
    js = json.loads(r.to_json(normalize=True))
    assert js[0]["name"] == "box" and abs(js[0]["box"]["x2"] - 110 / 160) < 1e-5
    # to_df / to_csv / to_xml / tojson (reference results.py:354-418): the summary as a table, through pandas
    df = r.to_df()
    assert list(df.columns) == ["name", "class", "confidence", "box"] and list(df["name"]) == ["box", "cat"]
    assert df["box"][0] == {"x1": 50.0, "y1": 40.0, "x2": 110.0, "y2": 80.0} and abs(df["confidence"][1] - 0.4) < 1e-5
    csv = r.to_csv(normalize=True, index=False)
    assert csv.splitlines()[0] == "name,class,confidence,box" and csv.splitlines()[1].startswith("box,1,0.9,")
    xml = r.to_xml()
    assert xml.startswith("<?xml") and "<name>box</name>" in xml and "<class>1</class>" in xml
    empty = Results(img, path="p", names={0: "a"}, boxes=torch.zeros((0, 6)))
    assert empty.to_xml() == '<?xml version="1.0" encoding="utf-8"?>\n<root></root>' and empty.to_df().empty
    assert json.loads(r.tojson()) == json.loads(r.to_json())  # (deprecated spelling, warns)
    with pytest.raises(ValueError):
        Results(None, path="p", names={0: "a"}, boxes=boxes[:1], orig_shape=(120, 160)).plot()


def test_uint8_scale_by_reciprocal_is_exact_in_bf16():
    """csrc/stem2.hip converts uint8 pixels with byte * (1 / 255.f) instead of the reference's byte / 255
    (predictor.py:84): the fp32 products differ in the last bit for about half of the byte values, but after the
    rounding to bf16 that follows immediately all 256 agree -- checked exhaustively here."""
    import numpy as np

    x = np.arange(256, dtype=np.float32)
    div = torch.from_numpy((x / np.float32(255.0)).astype(np.float32)).to(torch.bfloat16)
    mul = torch.from_numpy((x * np.float32(1.0 / 255.0)).astype(np.float32)).to(torch.bfloat16)
    assert torch.equal(div, mul)


def test_tensor_batches_loader():
    """LoadTensorBatches: a sequence of BCHW tensors, each one batch, with running image names; per-element checks as LoadTensor."""
    from yololite_amd.data import load_inference_source
    from yololite_amd.data.loaders import LoadTensorBatches

    ds = load_inference_source(LoadTensorBatches([torch.rand(2, 3, 32, 64), torch.rand(2, 3, 32, 64), torch.rand(1, 3, 32, 64)]))
    assert ds.source_type.tensor and len(ds) == 3 and not getattr(ds, "single_batch", False)
    got = list(ds)
    assert [p for b in got for p in b[0]] == [f"image{i}.jpg" for i in range(5)]
    assert got[2][1].shape == (1, 3, 32, 64)
    with pytest.raises(ValueError):
        LoadTensorBatches([torch.rand(1, 3, 30, 64)])
    with pytest.raises(ValueError):
        LoadTensorBatches([])


def test_image_file_loader_walks_files_dirs_globs_and_lists(tmp_path):
    """LoadImagesAndVideos (reference loaders.py:268-330): a directory, a glob and a .txt list give sorted files, batches of
    ``batch`` with a ragged last one, the reference's log prefixes; video files raise; unreadable images raise."""
    from PIL import Image

    for k in range(3):
        Image.fromarray(np.full((8, 10, 3), 40 * k, np.uint8)).save(tmp_path / f"im{k}.jpg")
    (tmp_path / "list.txt").write_text("\n".join(str(tmp_path / f"im{k}.jpg") for k in (2, 0)))
    batches = list(LoadImagesAndVideos(str(tmp_path), batch=2))
    assert [len(b[0]) for b in batches] == [2, 1] and len(LoadImagesAndVideos(str(tmp_path), batch=2)) == 2
    assert [Path(p).name for b in batches for p in b[0]] == ["im0.jpg", "im1.jpg", "im2.jpg"]
    assert batches[0][1][0].shape == (8, 10, 3) and batches[0][1][0].dtype == np.uint8
    assert batches[1][2][0].startswith("image 3/3 ")
    assert [Path(p).name for p in next(iter(LoadImagesAndVideos(str(tmp_path / "*.jpg"), batch=8)))[0]] == ["im0.jpg", "im1.jpg", "im2.jpg"]
    assert [Path(p).name for p in next(iter(LoadImagesAndVideos(str(tmp_path / "list.txt"), batch=8)))[0]] == ["im0.jpg", "im2.jpg"]
    (tmp_path / "clip.mp4").write_bytes(b"x")
    with pytest.raises(NotImplementedError):
        LoadImagesAndVideos(str(tmp_path))
    (tmp_path / "clip.mp4").unlink()
    (tmp_path / "bad.png").write_bytes(b"not an image")
    with pytest.raises(FileNotFoundError):
        list(LoadImagesAndVideos(str(tmp_path / "bad.png")))


def test_comm_id_file_ignores_another_jobs_id(tmp_path):
    """NativeComm.id_via_file: a file left behind by an earlier job (or written by another job's faster rank 0) carries another job
    token and is never taken for this job's id; this job's id is found once its rank 0 has written it; rank 0 removes the file."""
    import threading
    import time

    from yololite_amd.dist import NativeComm

    path = str(tmp_path / "id")
    stale = NativeComm.id_via_file(path, token="job-A")
    assert stale(b"A" * 128) == b"A" * 128            # an earlier job's rank 0 wrote its id and never cleaned up
    reader = NativeComm.id_via_file(path, token="job-B", timeout_s=0.3)
    with pytest.raises(TimeoutError):
        reader(None)                                   # the stale id is NOT returned
    reader = NativeComm.id_via_file(path, token="job-B", timeout_s=10.0)
    got = []
    t = threading.Thread(target=lambda: got.append(reader(None)))
    t.start()
    time.sleep(0.2)
    writer = NativeComm.id_via_file(path, token="job-B")
    writer(b"B" * 128)
    t.join(10)
    assert got == [b"B" * 128]
    writer.done(1)
    assert (tmp_path / "id").exists()                  # only rank 0 removes it
    writer.done(0)
    assert not (tmp_path / "id").exists()
    with pytest.raises(ValueError):
        NativeComm.id_via_file(path, token="")
