#!/usr/bin/env python3
"""Development aid: where does the fused stem+conv output differ from the two-launch form?  usage: H W"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from yololite_amd.nn import arch
from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.lower import lower
from yololite_amd.nn.synth import synth_state_dict

H, W = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
specs, _ = arch.parse_model(arch.yolo11_dict("n"))
sd = synth_state_dict("n", 4)
x = torch.randint(0, 256, (2, 3, H, W), generator=torch.Generator().manual_seed(1), dtype=torch.uint8).to(dev)
outs = []
for fuse in ("1", "0"):
    os.environ["YL_FUSE_STEM"] = fuse
    prog = lower(specs, sd, 2, H, W)
    eng = HipEngine(prog, dev)
    eng.forward(x)
    torch.cuda.synchronize()
    v = prog.ops[0]["out"] if fuse == "1" else prog.ops[1]["out"]
    print(fuse, prog.ops[0]["name"], eng.op_kernel(0), v.H, v.W, v.C)
    outs.append(eng.view_tensor(v).cpu())
a, b = outs
d = (a != b)
print("mismatches", int(d.sum()), "of", d.numel())
if d.any():
    idx = d.nonzero()
    print("batch", idx[:, 0].unique().tolist(), "chan", idx[:, 1].unique().tolist()[:40])
    print("rows", idx[:, 2].unique().tolist(), "cols", idx[:, 3].unique().tolist())
    print("max abs diff", float((a - b).abs().max()))
