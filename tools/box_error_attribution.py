"""Where the box error of the bf16 program comes from (CPU, no GPU needed): the oracle's bf16-storage emulation against the
reference-made goldens (tests/golden/golden_v1.npz) with the box branch's roundings switched off step by step.

    python tools/box_error_attribution.py

  all bf16, fp32 DFL logits   the product since round 6 (the decode runs on the fp32 accumulators of cv2[i][2])
  box branch unrounded        ... and cv2[i][0], cv2[i][1] keep fp32 outputs (bf16 filters)
  box branch fp32             ... and fp32 filters: only the backbone / neck are bf16
  only the box branch bf16    everything else fp32
The review of round 5 expected <= 0.5 px from moving the decode onto fp32 accumulators ("the whole source of the 0.2-0.3 stride box
error").  It is not: with an entirely fp32 box branch on bf16 FEATURES the maxima stay at 1.8-2.1 px on these random-init networks
-- the error is the bf16 storage of the activations upstream, which the DFL expectation of a synthetic (flat) logit profile
amplifies.  The fused decode still removes the logits' own rounding and 112 of 128 bytes per anchor.
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import yolo11_oracle as oracle  # noqa: E402
from yololite_amd.nn.synth import synth_state_dict  # noqa: E402

G = np.load(ROOT / "tests" / "golden" / "golden_v1.npz")
orig_conv = oracle.conv_bn_act
MODE = [None]


def conv(sd, p, x, *a, **k):
    m, box = MODE[0], (".cv2." in p and p.startswith("model.23"))
    if (m == "box_fp32" and box) or (m == "only_box" and not box):
        prev = oracle._STORAGE[0]
        oracle._STORAGE[0] = "fp32"
        try:
            return orig_conv(sd, p, x, *a, **k)
        finally:
            oracle._STORAGE[0] = prev
    if m == "box_unrounded" and box:
        w, b = oracle.fuse_conv_bn(sd[p + ".conv.weight"], sd[p + ".bn.weight"], sd[p + ".bn.bias"], sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"])
        kk = a[0] if a else k.get("k", 1)
        return F.silu(F.conv2d(x, w.bfloat16().float(), b, 1, kk // 2))
    return orig_conv(sd, p, x, *a, **k)


oracle.conv_bn_act = conv
for scale, h, w, seed in (("n", 64, 64, 0), ("n", 96, 128, 1), ("s", 64, 64, 2)):
    tag = f"net_{scale}_{h}x{w}"
    sd = synth_state_dict(scale, seed, cls_bias=list(G[tag + "_bias"]), cls_gain=8.0)
    x, ref = torch.from_numpy(G[tag + "_x"]), torch.from_numpy(G[tag + "_y"])
    st = torch.tensor([float(s) for s in (8, 16, 32) for _ in range((h // s) * (w // s))])
    with torch.inference_mode():
        yb, _ = oracle.forward(sd, x, scale, storage="bf16", box_logits="bf16")
        d = (yb[:, :4] - ref[:, :4]).abs()
        print(f"{tag:14s} {'bf16 DFL logits (rounds 1-5)':28s} max {float(d.max()):.3f} px  per level {[round(float(d[:, :, st == s].max()), 3) for s in (8, 16, 32)]}  rms {float(d.pow(2).mean().sqrt()):.4f}")
        for m, label in ((None, "all bf16, fp32 DFL logits"), ("box_unrounded", "box branch unrounded"), ("box_fp32", "box branch fp32"),
                         ("only_box", "only the box branch bf16")):
            MODE[0] = m
            y, _ = oracle.forward(sd, x, scale, storage="bf16", box_logits="fp32")
            d = (y[:, :4] - ref[:, :4]).abs()
            print(f"{tag:14s} {label:28s} max {float(d.max()):.3f} px  per level {[round(float(d[:, :, st == s].max()), 3) for s in (8, 16, 32)]}  rms {float(d.pow(2).mean().sqrt()):.4f}")
        MODE[0] = None
