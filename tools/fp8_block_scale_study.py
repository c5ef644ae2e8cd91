#!/usr/bin/env python3
"""CPU study (no GPU, no kernel): would CDNA4's E8M0 block scales (one power-of-two scale per 32 K-elements, the MX format
v_mfma_scale_f32_16x16x128_f8f6f4 consumes) let the 1x1 convs run in e4m3 inside the parity gate?

Round 1 found that e4m3 inputs for EVERY conv (3x3 and 1x1, per-tensor static scales) put the level maps 1.6-3.1 % off the
bf16 program and failed the post-NMS gate, while e4m3 on the dense 3x3 convs alone stays at 0.7-0.9 % and passes.  The
review suggested block scales.  This script measures, with the oracle (fp32 arithmetic), what the INPUT quantisation of
the convs costs under three schemes and which convs it is applied to:

  tensor : e4m3(x / s), s = absmax(tensor) * 2 / 448            (what conv_f8.hip does today)
  block  : e4m3(x / 2^e), e = ceil(log2(absmax(block) / 448)), block = 32 consecutive channels of one pixel (MX)
  bf16   : no fp8 (the reference point: bf16 storage only)

Usage: python tools/fp8_block_scale_study.py [scale] [H] [W]      -> prints relative L2 errors of the level maps vs fp32
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import yolo11_oracle as oracle  # noqa: E402
from yololite_amd.nn.synth import synth_state_dict  # noqa: E402

E4M3_MAX = 448.0


def q_tensor(x):
    s = x.abs().amax().clamp(min=1e-12) * 2.0 / E4M3_MAX
    return (x / s).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() * s


def q_block(x):  # x: (B, C, H, W); blocks of 32 channels per pixel, E8M0 (power-of-two) scales
    B, C, H, W = x.shape
    pad = (-C) % 32
    xp = F.pad(x, (0, 0, 0, 0, 0, pad)) if pad else x
    xb = xp.view(B, (C + pad) // 32, 32, H, W)
    m = xb.abs().amax(2, keepdim=True).clamp(min=2.0 ** -120)
    s = torch.exp2(torch.ceil(torch.log2(m / E4M3_MAX)))
    y = ((xb / s).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() * s).view(B, C + pad, H, W)
    return y[:, :C]


def run(sd, x, scale, scheme, which):
    """which: '3x3' (dense 3x3 only), 'all' (dense 3x3 and 1x1)."""
    orig = oracle.conv_bn_act

    def patched(sd_, p, xin, k=1, s=1, g=1, act=True, res=None):
        if scheme != "bf16" and g == 1 and xin.shape[1] >= 32 and (k == 3 or which == "all"):
            xin = q_tensor(xin) if scheme == "tensor" else q_block(xin)
        return orig(sd_, p, xin, k, s, g, act, res)

    oracle.conv_bn_act = patched
    try:
        with torch.inference_mode():
            return oracle.forward(sd, x, scale, storage="bf16")[1]
    finally:
        oracle.conv_bn_act = orig


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "n"
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 320
    torch.set_num_threads(8)
    sd = synth_state_dict(scale, 0, cls_bias=[-4.0, -4.0, -4.0], cls_gain=8.0)
    x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(5))
    with torch.inference_mode():
        ref = oracle.forward(sd, x, scale)[1]
    print(f"YOLO11{scale} {H}x{W}: relative L2 error of the level maps (P3, P4, P5) against the fp32 oracle")
    for scheme, which in (("bf16", "-"), ("tensor", "3x3"), ("block", "3x3"), ("tensor", "all"), ("block", "all")):
        lv = run(sd, x, scale, scheme, which)
        print(f"  {scheme:6s} fp8 inputs on {which:3s}:", [round(float((a - b).norm() / b.norm()), 4) for a, b in zip(lv, ref)])


if __name__ == "__main__":
    main()
