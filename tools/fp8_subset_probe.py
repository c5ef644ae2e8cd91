#!/usr/bin/env python3
"""Which e4m3 convs cost the detections their accuracy?  YOLO11x 1280x1280 batch 2: the e4m3 program with SUBSETS of the dense
3x3 convs in e4m3 (the others bf16) against the bf16 program: boxes kept at conf 0.25, class-map error, time per batch.
Usage: python tools/fp8_subset_probe.py [batch]"""
import re
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench
from yololite_amd.nn import arch
from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.lower import lower

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = 1280
x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(0)).to(dev)
x8 = (x * 255).to(torch.uint8)
specs, sd, e16 = bench.build_model("x", dev, B, S, S, x, target_frac=bench.config5_target_frac(S), fp8=False)
o16, c16 = e16.detect(x8, 0.25, 0.7, graph=False)
l16 = [t.clone() for t in e16.raw_cls_levels()]
scales = e16.calibrate_fp8(x, 2.0)
torch.cuda.synchronize()


def timeit(eng, n=10):
    eng.detect(x8, 0.25, 0.7, graph=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        eng.detect(x8, 0.25, 0.7, graph=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"bf16: boxes {c16.tolist()}  {timeit(e16):.2f} ms")
names = sorted(scales)
layer = lambda n: int(re.match(r"model\.(\d+)", n).group(1))
from oracle.parity_gate import parity_gate

ref16 = [o16[i, : int(c16[i])].cpu() for i in range(B)]
subsets = {
    "all": names,
    "5-23": [n for n in names if layer(n) >= 5],
    "3-23": [n for n in names if layer(n) >= 3],
    "7-23": [n for n in names if layer(n) >= 7],
    "5-22": [n for n in names if 5 <= layer(n) <= 22],
    "backbone 0-10": [n for n in names if layer(n) <= 10],
    "0-8": [n for n in names if layer(n) <= 8],
    "0-6": [n for n in names if layer(n) <= 6],
    "0-4": [n for n in names if layer(n) <= 4],
    "no head (0-22)": [n for n in names if layer(n) <= 22],
    "neck+head 11-23": [n for n in names if layer(n) >= 11],
    "head only 23": [n for n in names if layer(n) == 23],
}
for tag, sub in subsets.items():
    sc = {n: scales[n] for n in sub}
    eng = HipEngine(lower(specs, sd, B, S, S, fp8_scales=sc), dev)
    o, c = eng.detect(x8, 0.25, 0.7, graph=False)
    torch.cuda.synchronize()
    errs = [float((a - b).norm() / b.norm()) for a, b in zip(eng.raw_cls_levels(), l16)]
    nf8 = sum("_f8_" in eng.op_kernel(i) for i in range(len(eng.prog.ops)))
    g = parity_gate(ref16, [o[i, : int(c[i])].cpu() for i in range(B)], 0.25, 0.7, band=0.12, min_matched_frac=0.85, max_explained_share=0.35)
    print(f"{tag:18s} convs {len(sub):3d} (f8 launches {nf8:3d})  boxes {c.tolist()}  class-map rel-L2 {[round(e, 4) for e in errs]}  {timeit(eng):.2f} ms"
          f"  vs bf16: matched {g['matched_frac']:.3f} max_dconf {g['max_dconf']:.3f} explained_share {g['explained_share']:.3f} count_bias {g['count_bias']:+.3f}")
    del eng
    torch.cuda.empty_cache()
