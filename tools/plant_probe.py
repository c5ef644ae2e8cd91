"""GPU box: the planted-object head (nn/synth.py plant_object_head, bench.build_model(planted=...)) -- per-anchor best-score
histogram and the post-NMS gate of the HIP bf16 / e4m3 programs and of the CPU emulations against the fp32 oracle.

    python tools/plant_probe.py x 1280 2 [fp8]
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.argv, argv = sys.argv[:1], sys.argv[1:]
import bench  # noqa: E402
from oracle import yolo11_oracle as oracle  # noqa: E402
from oracle.parity_gate import parity_gate  # noqa: E402


def main(scale="x", size=1280, batch=2, dtype="fp8", planted=96, alpha=12.0, margin=3.0, ridge=0.05, ridge_samples=0.25):
    size, batch = int(size), int(batch)
    planted = dict(per_image=int(planted), alpha=float(alpha), margin=float(margin), ridge=float(ridge), ridge_samples=float(ridge_samples))
    print("planted head:", planted)
    dev = torch.device("cuda:0")
    x = bench.config5_images(batch, size).to(dev)
    x8 = (x * 255).to(torch.uint8)
    t0 = time.time()
    specs, sd, eng = bench.build_model(scale, dev, batch, size, size, x, fp8=dtype == "fp8", planted=planted)
    print("build", round(time.time() - t0, 1), "s")
    conf, iou = 0.25, 0.7
    o, c = eng.detect(x8, conf, iou, max_det=3000, graph=False)
    y = eng.forward(x8, graph=False).float().cpu()
    n_or = min(2, batch)
    xin = x8[:n_or].float().cpu() / 255.0
    torch.set_num_threads(8)
    with torch.inference_mode():
        y32, _ = oracle.forward(sd, xin, scale)
        ref = oracle.non_max_suppression(y32, conf, iou, max_det=3000)
    mx = y32[:, 4:].amax(1)
    bins = [0, 0.01, 0.05, 0.1, 0.2, 0.25, 0.3, 0.4, 0.6, 0.8, 0.9, 1.0]
    print("fp32 best-score histogram", bins, np.histogram(mx.numpy().ravel(), bins=bins)[0].tolist(), "over conf per image", (mx > conf).sum(1).tolist())
    keys = ("ref", "det", "matched_frac", "class_mismatch", "unexplained", "max_dconf", "p99_dconf", "explained_share", "count_bias", "near_threshold", "nms_flip", "ok")
    kw = dict(band=0.2, min_matched_frac=0.85, max_explained_share=0.35, ref_y=y32)
    g = parity_gate(ref, [o[i, : int(c[i])].cpu() for i in range(n_or)], conf, iou, got_y=y[:n_or], **kw)
    print(f"HIP {dtype} vs fp32 oracle:", {k: g.get(k) for k in keys}, "max |dscore| pre-NMS", round(float((y[:n_or, 4:] - y32[:, 4:]).abs().max()), 4))
    with torch.inference_mode():
        ye, _ = oracle.forward(sd, xin, scale, storage="bf16", fp8_scales=eng.fp8_conv_scales() if dtype == "fp8" else None)
        dete = oracle.non_max_suppression(ye, conf, iou, max_det=3000)
    ge = parity_gate(ref, dete, conf, iou, got_y=ye, **kw)
    print("emulation vs fp32 oracle:  ", {k: ge.get(k) for k in keys}, "max |dscore| pre-NMS", round(float((ye[:, 4:] - y32[:, 4:]).abs().max()), 4))
    # the anchors where the HIP program is farthest from fp32: level, position (borders?), class, the three scores
    d = (y[:n_or, 4:] - y32[:, 4:]).abs()
    top = torch.topk(d.flatten(), 8).indices.tolist()
    A = y32.shape[2]
    lv, a0 = [], 0
    for st in (8, 16, 32):
        lv.append((a0, size // st))
        a0 += (size // st) ** 2
    for t in top:
        b, c, a = t // (80 * A), (t // A) % 80, t % A
        l = max(i for i, (s0, _) in enumerate(lv) if a >= s0)
        py, px = divmod(a - lv[l][0], lv[l][1])
        print(f"  worst: image {b} level {l} ({py}, {px}) of {lv[l][1]} class {c}: fp32 {float(y32[b, 4 + c, a]):.4f} HIP {float(y[b, 4 + c, a]):.4f} emulation {float(ye[b, 4 + c, a]):.4f}"
              f" | anchor best fp32 {float(y32[b, 4:, a].max()):.4f}")
    if dtype == "fp8":
        from yololite_amd.nn.engine import HipEngine

        e16 = HipEngine.from_model(specs, sd, batch, size, size, dev)
        o16, c16 = e16.detect(x8, conf, iou, max_det=3000, graph=False)
        y16 = e16.forward(x8, graph=False).float().cpu()
        g16 = parity_gate(ref, [o16[i, : int(c16[i])].cpu() for i in range(n_or)], conf, iou, got_y=y16[:n_or], **kw)
        print("HIP bf16 vs fp32 oracle:   ", {k: g16.get(k) for k in keys}, "max |dscore| pre-NMS", round(float((y16[:n_or, 4:] - y32[:, 4:]).abs().max()), 4))


if __name__ == "__main__":
    main(*argv)
