#!/usr/bin/env python3
"""White-noise inputs against structured synthetic images (nn/synth.py::synth_images): spread of the per-anchor maximum class
logit, and how the box count at conf 0.25 moves between the fp32 oracle, the bf16 program and the e4m3 program."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench
from oracle import yolo11_oracle as oracle
from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.synth import synth_images

dev = torch.device("cuda:0")
scale, B, S = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("n", 4, 640)
for kind in ("noise", "structured"):
    x = (torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(0)) if kind == "noise" else synth_images(B, S, S, 0)).to(dev)
    x8 = (x * 255).to(torch.uint8)
    specs, sd, e16 = bench.build_model(scale, dev, B, S, S, x, target_frac=bench.config5_target_frac(S))
    e16.forward(x8)
    mx = torch.cat([r.amax(1).float().flatten(1) for r in e16.raw_cls_levels()], 1)
    o16, c16 = e16.detect(x8, 0.25, 0.7, graph=False)
    e8 = HipEngine.from_model(specs, sd, B, S, S, dev, fp8_calib=x)
    o8, c8 = e8.detect(x8, 0.25, 0.7, graph=False)
    torch.cuda.synchronize()
    torch.set_num_threads(8)
    n_or = min(B, 2)
    with torch.inference_mode():
        y32, _ = oracle.forward(sd, x8[:n_or].float().cpu() / 255.0, scale)
        ref = oracle.non_max_suppression(y32, 0.25, 0.7)
    q = torch.quantile(mx.flatten().cpu(), torch.tensor([0.5, 0.9, 0.985, 0.996, 0.999]))
    print(f"{kind:10s} {scale}@{S}: max-logit over anchors: std {float(mx.std()):.3f}, quantiles 50/90/98.5/99.6/99.9 % {[round(float(v), 3) for v in q]}")
    print(f"           boxes fp32 {[len(r) for r in ref]}  bf16 {c16.tolist()[:n_or]}  e4m3 {c8.tolist()[:n_or]}")
    del e16, e8
    torch.cuda.empty_cache()
