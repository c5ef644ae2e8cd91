"""Generate golden vectors by importing the REFERENCE (runs only in the build container).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

/root/reference is imported with two import-time stand-ins for third-party modules that are absent
from the image: ``cv2`` (only ``setNumThreads`` and constants are touched at
import; ``resize``/``copyMakeBorder`` record their arguments so LetterBox *geometry* can be pinned)
and ``torchvision`` whose ``ops.nms`` is our restatement of the published greedy NMS (so the F3
vectors pin the reference's pre/post-processing around NMS, not torchvision itself).
Weights come from this repo's deterministic synthesizer and are loaded into the reference module via
``load_state_dict``; nothing of the reference is written to the fixtures except numbers it computed.
"""
from __future__ import annotations

import hashlib
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True
os.environ["YOLO_OFFLINE"] = "true"
os.environ["YOLO_CONFIG_DIR"] = "/tmp/yl_golden_cfg"
os.makedirs("/tmp/yl_golden_cfg", exist_ok=True)

from oracle import yolo11_oracle as oracle  # noqa: E402  (used only for the torchvision stand-in)

CV2_CALLS = []


class _CV2(types.ModuleType):
    IMREAD_COLOR = 1
    INTER_LINEAR = 1
    BORDER_CONSTANT = 0

    def setNumThreads(self, n):
        pass

    def resize(self, img, size, interpolation=None):
        CV2_CALLS.append(("resize", tuple(size)))
        return np.zeros((size[1], size[0], img.shape[2]), dtype=img.dtype)

    def copyMakeBorder(self, img, top, bottom, left, right, kind, value=None):
        CV2_CALLS.append(("border", (top, bottom, left, right), tuple(value)))
        return np.zeros((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), dtype=img.dtype)

    def __getattr__(self, k):
        def f(*a, **kw):
            raise RuntimeError("cv2 stand-in: " + k)
        return f


def _nms(boxes, scores, iou):
    keep = oracle.nms_greedy(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), float(iou))
    return torch.from_numpy(keep).to(boxes.device)


sys.modules["cv2"] = _CV2("cv2")
_tv = types.ModuleType("torchvision")
_tv.__version__ = "0.0-standin"
_tv.ops = types.ModuleType("torchvision.ops")
_tv.ops.nms = _nms
sys.modules["torchvision"] = _tv
sys.modules["torchvision.ops"] = _tv.ops
sys.path.insert(0, "/root/reference")

from yololite.data.augment import LetterBox  # noqa: E402
from yololite.nn.tasks import DetectionModel, yaml_model_load  # noqa: E402
from yololite.utils import ops as ref_ops  # noqa: E402

from yololite_amd.nn.synth import calibrate_cls_bias, synth_state_dict  # noqa: E402


def sd_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def ref_model(scale, sd):
    m = DetectionModel(yaml_model_load(f"/root/reference/yololite/cfg/yolo11{scale}.yaml"), verbose=False)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.eval()


def rand_input(seed, b, h, w):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, 3, h, w, generator=g)


def calibrated_sd(scale, seed, x, gain=8.0, frac=0.015):
    """Class bias calibrated with the reference forward so about `frac` of anchors pass conf 0.25."""
    sd0 = synth_state_dict(scale, seed, cls_bias=[0.0, 0.0, 0.0], cls_gain=gain)
    m = ref_model(scale, sd0)
    with torch.inference_mode():
        _, feats = m(x)
    mx = [f[:, 64:].amax(1) for f in feats]
    bias = calibrate_cls_bias(mx, frac)
    return synth_state_dict(scale, seed, cls_bias=bias, cls_gain=gain), bias


def main():
    out = {}
    torch.manual_seed(0)
    torch.set_num_threads(4)

    # ---------------- F2: whole-network outputs at small sizes (full tensors)
    for scale, b, h, w, seed in (("n", 2, 64, 64, 0), ("n", 2, 96, 128, 1), ("s", 1, 64, 64, 2), ("m", 1, 64, 64, 3),
                                 ("x", 1, 64, 64, 4)):
        x = rand_input(100 + seed, b, h, w)
        sd, bias = calibrated_sd(scale, seed, x)
        m = ref_model(scale, sd)
        with torch.inference_mode():
            y, feats = m(x.clone())
        tag = f"net_{scale}_{h}x{w}"
        out[tag + "_x"] = x.numpy()
        out[tag + "_y"] = y.numpy()
        for i, f in enumerate(feats):
            out[f"{tag}_raw{i}"] = f.numpy()
        out[tag + "_bias"] = np.asarray(bias, dtype=np.float64)
        out[tag + "_sd"] = np.frombuffer(sd_digest(sd).encode(), dtype=np.uint8)
        out[tag + "_meta"] = np.asarray([b, h, w, seed], dtype=np.int64)
        print(tag, y.shape, "cands/img", (y[:, 4:].amax(1) > 0.25).sum(1).tolist())

    # ---------------- F1: every layer output for n at 64x64 (B=1)
    x = rand_input(200, 1, 64, 64)
    sd, bias = calibrated_sd("n", 0, x)
    m = ref_model("n", sd)
    layer_out = {}
    hooks = [mod.register_forward_hook(lambda mod_, inp, o, i=i: layer_out.__setitem__(i, o))
             for i, mod in enumerate(m.model)]
    with torch.inference_mode():
        y, feats = m(x.clone())
    for hk in hooks:
        hk.remove()
    out["layers_n_x"] = x.numpy()
    out["layers_n_bias"] = np.asarray(bias)
    for i in range(23):
        out[f"layers_n_{i}"] = layer_out[i].numpy()
    out["layers_n_y"] = y.numpy()

    # ---------------- F2b: 640x640, n, B=1: strided subsample + per-channel sums
    x = rand_input(300, 1, 640, 640)
    sd, bias = calibrated_sd("n", 0, x)
    m = ref_model("n", sd)
    with torch.inference_mode():
        y, feats = m(x.clone())
    out["net_n_640_bias"] = np.asarray(bias)
    out["net_n_640_ysub"] = y[:, :, ::37].numpy()
    out["net_n_640_ysum"] = y.double().sum(2).numpy()
    out["net_n_640_yabsmax"] = y.abs().amax(2).numpy()
    out["net_n_640_xsum"] = np.asarray([x.double().sum().item()])
    dets = ref_ops.non_max_suppression(y.clone(), 0.25, 0.7, max_det=300)
    out["net_n_640_det0"] = dets[0].numpy()
    print("640 det", dets[0].shape, "cands", int((y[:, 4:].amax(1) > 0.25).sum()))

    # ---------------- F3: planted NMS cases through the reference's non_max_suppression
    def planted(seed, b, a, nc, n_obj, jitter, same_box_two_classes=False, score_lo=0.3):
        g = torch.Generator().manual_seed(seed)
        p = torch.zeros(b, 4 + nc, a)
        p[:, 4:] = torch.rand(b, nc, a, generator=g) * 0.2  # background below threshold
        for bi in range(b):
            centers = torch.rand(n_obj, 2, generator=g) * 500 + 60
            sizes = torch.rand(n_obj, 2, generator=g) * 120 + 20
            idx = torch.randperm(a, generator=g)
            k = 0
            for o in range(n_obj):
                reps = int(torch.randint(1, 6, (1,), generator=g))
                cls = int(torch.randint(0, nc, (1,), generator=g))
                for r in range(reps):
                    ai = idx[k]
                    k += 1
                    p[bi, 0:2, ai] = centers[o] + (torch.rand(2, generator=g) - 0.5) * jitter
                    p[bi, 2:4, ai] = sizes[o] * (1 + (torch.rand(2, generator=g) - 0.5) * jitter / 40)
                    c2 = cls if not (same_box_two_classes and r % 2) else (cls + 1) % nc
                    p[bi, 4 + c2, ai] = score_lo + (1 - score_lo) * torch.rand(1, generator=g)
        return p

    cases = {
        "sep": dict(pred=planted(1, 2, 512, 80, 12, 4.0), kw=dict(conf_thres=0.25, iou_thres=0.7)),
        "twocls": dict(pred=planted(2, 2, 512, 80, 10, 2.0, True), kw=dict(conf_thres=0.25, iou_thres=0.7)),
        "hiiou": dict(pred=planted(3, 1, 800, 80, 20, 30.0), kw=dict(conf_thres=0.25, iou_thres=0.45)),
        "agnostic": dict(pred=planted(4, 2, 512, 80, 10, 2.0, True), kw=dict(conf_thres=0.25, iou_thres=0.7, agnostic=True)),
        "classes": dict(pred=planted(5, 2, 512, 20, 30, 6.0), kw=dict(conf_thres=0.25, iou_thres=0.7, classes=[0, 3, 7])),
        "cls79": dict(pred=planted(6, 1, 600, 80, 15, 10.0), kw=dict(conf_thres=0.3, iou_thres=0.6)),
        "empty": dict(pred=planted(7, 2, 256, 80, 0, 0.0), kw=dict(conf_thres=0.25, iou_thres=0.7)),
    }
    # many survivors: a grid of disjoint boxes, > max_det of them, to exercise [:max_det]
    a = 1024
    p = torch.zeros(1, 84, a)
    g = torch.Generator().manual_seed(8)
    gx, gy = torch.meshgrid(torch.arange(32.0), torch.arange(32.0), indexing="ij")
    p[0, 0] = gx.flatten() * 20 + 10
    p[0, 1] = gy.flatten() * 20 + 10
    p[0, 2:4] = 12.0
    p[0, 4 + 79] = 0.3 + 0.6 * torch.rand(a, generator=g)
    cases["trunc"] = dict(pred=p, kw=dict(conf_thres=0.25, iou_thres=0.7, max_det=300))
    # force class 79 boxes (offset 606720: fp32 ulp 1/16 px) in the cls79 case
    pc = cases["cls79"]["pred"]
    sc = pc[:, 4:].clone()
    mx = sc.amax(1, keepdim=True)
    pc[:, 4:] = 0
    pc[:, 4 + 79] = mx[:, 0]
    for name, c in cases.items():
        res = ref_ops.non_max_suppression(c["pred"].clone(), **c["kw"])
        out[f"nms_{name}_pred"] = c["pred"].numpy()
        out[f"nms_{name}_n"] = np.asarray([len(r) for r in res], dtype=np.int64)
        out[f"nms_{name}_out"] = np.concatenate([r.numpy() for r in res], 0) if res else np.zeros((0, 6), np.float32)
        kw = c["kw"]
        out[f"nms_{name}_kw"] = np.asarray([kw.get("conf_thres", 0.25), kw.get("iou_thres", 0.45),
                                            float(kw.get("agnostic", False)), kw.get("max_det", 300)], dtype=np.float64)
        out[f"nms_{name}_classes"] = np.asarray(kw.get("classes", []), dtype=np.int64)
        print("nms", name, out[f"nms_{name}_n"].tolist())

    # ---------------- F4: LetterBox geometry and scale_boxes
    shapes = [(1080, 1920), (480, 640), (640, 481), (500, 381), (427, 640), (640, 640), (333, 500), (720, 1280)]
    geo = []
    for (h0, w0) in shapes:
        for auto in (True, False):
            CV2_CALLS.clear()
            img = np.zeros((h0, w0, 3), dtype=np.uint8)
            o = LetterBox((640, 640), auto=auto, stride=32)(image=img)
            rs = [c for c in CV2_CALLS if c[0] == "resize"]
            bd = [c for c in CV2_CALLS if c[0] == "border"][0]
            new_unpad = rs[0][1] if rs else (w0, h0)
            geo.append([h0, w0, int(auto), new_unpad[0], new_unpad[1], *bd[1], o.shape[0], o.shape[1], bd[2][0]])
    out["letterbox_geo"] = np.asarray(geo, dtype=np.int64)
    sb_in, sb_out, sb_shapes = [], [], []
    g = torch.Generator().manual_seed(9)
    for row in geo:
        h0, w0, h1, w1 = row[0], row[1], row[9], row[10]
        boxes = torch.rand(6, 4, generator=g) * torch.tensor([w1, h1, w1, h1]) * 1.2 - 20
        boxes[0] = torch.tensor([10.0, 20.0, w1 - 10.0, h1 - 14.0])
        sb_in.append(boxes.numpy().copy())
        sb_out.append(ref_ops.scale_boxes((h1, w1), boxes.clone(), (h0, w0)).numpy())
        sb_shapes.append([h1, w1, h0, w0])
    out["scale_boxes_in"] = np.stack(sb_in)
    out["scale_boxes_out"] = np.stack(sb_out)
    out["scale_boxes_shapes"] = np.asarray(sb_shapes, dtype=np.int64)

    # known answers from the reference module itself
    m = DetectionModel(yaml_model_load("/root/reference/yololite/cfg/yolo11n.yaml"), verbose=False)
    out["ka_stride"] = m.stride.numpy()
    out["ka_save"] = np.asarray(m.save)
    out["ka_params"] = np.asarray([sum(p.numel() for p in DetectionModel(
        yaml_model_load(f"/root/reference/yololite/cfg/yolo11{s}.yaml"), verbose=False).parameters()) for s in "nsmlx"])
    out["ka_sd_keys"] = np.frombuffer("\n".join(m.state_dict().keys()).encode(), dtype=np.uint8)

    np.savez_compressed(HERE / "golden_v1.npz", **out)
    print("wrote", HERE / "golden_v1.npz", (HERE / "golden_v1.npz").stat().st_size / 1e6, "MB")


if __name__ == "__main__":
    main()
