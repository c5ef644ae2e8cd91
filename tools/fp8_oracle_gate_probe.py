#!/usr/bin/env python3
"""Development probe: YOLO11x 1280x1280, e4m3 and bf16 HIP programs against the fp32 oracle on ONE image (config 5's gate)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench
from oracle import yolo11_oracle as oracle
from oracle.parity_gate import parity_gate

dev = torch.device("cuda:0")
B, S = 8, 1280
x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(0)).to(dev)
x8 = (x * 255).to(torch.uint8)
for max_det in (300, 3000):
    res = {}
    for dtype in ("bf16", "fp8"):
        specs, sd, eng = bench.build_model("x", dev, B, S, S, x, fp8=dtype == "fp8")
        o, c = eng.detect(x8, 0.25, 0.7, max_det=max_det, graph=False)
        torch.cuda.synchronize()
        res[dtype] = [o[0, : int(c[0])].cpu()]
        del eng
        torch.cuda.empty_cache()
    torch.set_num_threads(8)
    t = time.time()
    with torch.inference_mode():
        y32, _ = oracle.forward(sd, x8[:1].float().cpu() / 255.0, "x")
    ref = oracle.non_max_suppression(y32, 0.25, 0.7, max_det=max_det)
    print("oracle forward + nms", round(time.time() - t, 1), "s; ref boxes", len(ref[0]))
    for dtype, band in (("bf16", 0.08), ("fp8", 0.12)):
        g = parity_gate(ref, res[dtype], 0.25, 0.7, band=band, min_matched_frac=0.85, ref_y=y32)
        print("max_det", max_det, dtype, g)
