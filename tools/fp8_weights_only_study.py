#!/usr/bin/env python3
"""CPU study (no GPU, no kernel): what would OCP e4m3 WEIGHTS ONLY cost on the 1x1 convs (bf16 activations, fp32
accumulation; the round-2 review's open question for config 5)?

The folded filter of every dense 1x1 conv is rounded to e4m3 with one fp32 scale per OUTPUT channel
(w_q[o] = e4m3(w[o] / s_o) * s_o, s_o = absmax(w[o]) / 448) instead of bf16; everything else is the oracle's bf16-storage
mode, i.e. what the bf16 HIP program computes.  Printed: relative L2 error of the three level maps against the fp32 oracle
for bf16 weights (the format's own error), e4m3 weights on the 1x1 convs, and for comparison e4m3 INPUTS on the dense 3x3
convs (what conv_f8.hip does today and what passes the post-NMS gate).

Usage: python tools/fp8_weights_only_study.py [scale] [H] [W]
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import yolo11_oracle as oracle  # noqa: E402  (development study, not a product path)
from yololite_amd.nn.synth import synth_state_dict  # noqa: E402

E4M3_MAX = 448.0


def q_weights_e4m3(w):
    s = w.flatten(1).abs().amax(1).clamp(min=1e-30).view(-1, 1, 1, 1) / E4M3_MAX
    return (w / s).to(torch.float8_e4m3fn).float() * s


def q_inputs_e4m3(x):
    s = x.abs().amax().clamp(min=1e-12) * 2.0 / E4M3_MAX
    return (x / s).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() * s


def run(sd, x, scale, w8_1x1, in8_3x3):
    orig = oracle.conv_bn_act

    def patched(sd_, p, xin, k=1, s=1, g=1, act=True, res=None, dw_taps_bf16=False):
        if g != 1 or not ((k == 1 and w8_1x1) or (k == 3 and in8_3x3 and xin.shape[1] >= 32)):
            return orig(sd_, p, xin, k, s, g, act, res, dw_taps_bf16)
        w, b = oracle.fuse_conv_bn(sd_[p + ".conv.weight"], sd_[p + ".bn.weight"], sd_[p + ".bn.bias"],
                                   sd_[p + ".bn.running_mean"], sd_[p + ".bn.running_var"])
        if k == 1:
            w = q_weights_e4m3(w)
        else:
            w, xin = w.bfloat16().float(), q_inputs_e4m3(xin)
        y = F.conv2d(xin, w, b, s, k // 2, 1, 1)
        y = F.silu(y) if act else y
        return oracle.q_store(y if res is None else y + res)

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
    for name, w8, in8 in (("bf16 weights everywhere (the bf16 program)", False, False),
                          ("e4m3 weights on the 1x1 convs", True, False),
                          ("e4m3 inputs on the dense 3x3 convs (config 5 today)", False, True),
                          ("both", True, True)):
        lv = run(sd, x, scale, w8, in8)
        print(f"  {name:55s}", [round(float((a - b).norm() / b.norm()), 4) for a, b in zip(lv, ref)])


if __name__ == "__main__":
    main()
