#!/usr/bin/env python3
"""Development micro-benchmark: one conv layer at batch 32, cold-ish (rotating buffers), with ablations.
Usage: YL_CONV_DBG=<flags> python tools/conv_micro.py cin cout k s H W [fp8]   (fp8: e4m3 input, conv_f8_kernel)"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.lower import BF16, FP8, OP_CONV, Program, _Arena

cin, cout, k, s, H, W = map(int, sys.argv[1:7])
f8 = len(sys.argv) > 7 and sys.argv[7] == "fp8"
B = int(os.environ.get("YL_MICRO_B", "32"))
dev = torch.device("cuda:0")
ar = _Arena(B)
prog = Program(B, 32, 32, 80)
prog.input_view = ar.alloc(32, 32, 8)
p = k // 2
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
NREP = 6  # rotate through distinct buffers so inputs come from HBM / MALL, not L2
w = (np.random.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)
b = np.zeros(cout, np.float32)
for r in range(NREP):
    x, y = ar.alloc(H, W, cin, FP8 if f8 else BF16), ar.alloc(Ho, Wo, cout)
    prog.ops.append(dict(kind=OP_CONV, name=f"c{r}", k=k, stride=s, act=int(os.environ.get("YL_MICRO_ACT", "1")), inp=x, out=y, res=None, weight=w, bias=b, in_scale=0.05))
prog.ws_bytes = ar.top + 256
eng = HipEngine(prog, dev)
for o in prog.ops:
    eng.write_view(o['inp'], (torch.randn(B, cin, H, W) * (20 if f8 else 1)).clamp(-448, 448))
im = torch.zeros(B, 3, 32, 32, device=dev)
for _ in range(3):
    eng.forward(im, decode=False, graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 20
for _ in range(N):
    eng.forward(im, decode=False, graph=True)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / (N * NREP) * 1e3
byt = B * (H * W * cin * (1 if f8 else 2) + Ho * Wo * cout * 2)
torch.cuda.synchronize()
fl = 2.0 * B * Ho * Wo * cout * cin * k * k
print(eng.op_kernel(0), end="  ")
print(f"{'fp8 ' if f8 else ''}dbg={os.environ.get('YL_CONV_DBG', '0'):>2s} {cin}->{cout} k{k} s{s} @{H}x{W}: {us:7.1f} us  {byt / us / 1e6:6.2f} TB/s  {fl / us / 1e6:7.1f} TF/s")
