"""Safe-tensor export / import (``nn/export.py``)."""
import numpy as np
import pytest
import torch

from yololite_amd import YOLOLite
from yololite_amd.nn import arch
from yololite_amd.nn.export import fold_state_dict, is_folded, load_safetensors, save_safetensors
from yololite_amd.nn.lower import fold_conv_bn, lower
from yololite_amd.nn.synth import synth_state_dict
from yololite_amd.nn.tasks import DetectionModel, attempt_load_one_weight


def _same_program(pa, pb):
    assert len(pa.ops) == len(pb.ops)
    for a, b in zip(pa.ops, pb.ops):
        assert a["kind"] == b["kind"] and a.get("name") == b.get("name")
        for k in ("weight", "bias", "post_weight", "post_bias", "stem_weight", "stem_bias"):
            if a.get(k) is not None:
                assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (a.get("name"), k)


@pytest.mark.parametrize("folded", [True, False])
def test_safetensors_round_trip_gives_the_same_engine_program(tmp_path, folded):
    m = DetectionModel(arch.yolo11_dict("n"), verbose=False, seed=3)
    m.names = {i: f"cls{i}" for i in range(m.nc)}
    f = save_safetensors(m, tmp_path / "n.safetensors", folded=folded)
    m2, ckpt = attempt_load_one_weight(f)
    assert is_folded(m2.state_dict()) == folded
    assert m2.names == m.names and torch.equal(m2.stride, m.stride) and m2.yaml["scale"] == "n"
    assert m2.pt_path == str(f) and m2.task == "detect" and "imgsz" in m2.args
    # every packed filter / bias the engine would see is bit-identical
    _same_program(lower(m.specs, m.state_dict(), 2, 64, 64), lower(m2.specs, m2.state_dict(), 2, 64, 64))
    if folded:  # a Conv unit: folded once, stored, returned as stored
        w, b = fold_conv_bn(m.state_dict(), "model.3")
        w2, b2 = fold_conv_bn(m2.state_dict(), "model.3")
        assert np.array_equal(w, w2) and np.array_equal(b, b2)
        assert "model.3.bn.weight" not in m2.state_dict() and "model.23.cv2.0.2.bias" in m2.state_dict()


def test_facade_export_and_load(tmp_path):
    y = YOLOLite("yolo11n.yaml")
    f = y.export(file=tmp_path / "out" / "m.safetensors")
    y2 = YOLOLite(f)
    assert y2.ckpt_path == f and y2.task == "detect" and is_folded(y2.model.state_dict())
    with pytest.raises(NotImplementedError):
        y.export(format="onnx")


def test_foreign_and_mismatched_files_fail_loudly(tmp_path):
    from safetensors.torch import save_file

    save_file({"a": torch.zeros(2)}, str(tmp_path / "x.safetensors"))
    with pytest.raises(ValueError, match="not a yololite_amd"):
        load_safetensors(tmp_path / "x.safetensors")
    with pytest.raises(FileNotFoundError):
        load_safetensors(tmp_path / "missing.safetensors")
    m = DetectionModel(arch.yolo11_dict("n"), verbose=False)
    sd = fold_state_dict(m.state_dict())
    sd.pop("model.3.conv.bias")
    m._sd = sd
    save_safetensors(m, tmp_path / "broken.safetensors")
    with pytest.raises(RuntimeError, match="do not match"):
        load_safetensors(tmp_path / "broken.safetensors")


@pytest.mark.gpu
def test_engine_from_folded_file_is_bit_identical(gpu_device, tmp_path):
    from yololite_amd.nn.engine import HipEngine

    m = DetectionModel(arch.yolo11_dict("n"), verbose=False, state_dict=synth_state_dict("n", 5, cls_bias=[-2.0, -2.0, -2.0], cls_gain=4.0))
    m2, _ = load_safetensors(save_safetensors(m, tmp_path / "n.safetensors"))
    x = torch.randint(0, 256, (2, 3, 96, 128), generator=torch.Generator().manual_seed(1), dtype=torch.uint8).to(gpu_device)
    ya = HipEngine.from_model(m.specs, m.state_dict(), 2, 96, 128, gpu_device).forward(x).clone()
    yb = HipEngine.from_model(m2.specs, m2.state_dict(), 2, 96, 128, gpu_device).forward(x)
    assert torch.equal(ya, yb)
