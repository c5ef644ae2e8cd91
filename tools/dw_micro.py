#!/usr/bin/env python3
"""Development micro-benchmark: depthwise 3x3 at batch 32 inside a hipGraph (rotating buffers).
Usage: python tools/dw_micro.py C H W [Cout]   (Cout: fused depthwise + pointwise 1x1, dwpw_kernel)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.lower import OP_DWCONV3, Program, _Arena

C, H, W = map(int, sys.argv[1:4])
COUT = int(sys.argv[4]) if len(sys.argv) > 4 else 0
B = 32
dev = torch.device("cuda:0")
ar = _Arena(B)
prog = Program(B, 32, 32, 80)
prog.input_view = ar.alloc(32, 32, 8)
NREP = 6
w = (np.random.randn(C, 1, 3, 3) / 3).astype(np.float32)
b = np.zeros(C, np.float32)
for r in range(NREP):
    x, y = ar.alloc(H, W, C), ar.alloc(H, W, COUT or C)
    op = dict(kind=OP_DWCONV3, name=f"d{r}", act=1, inp=x, out=y, res=None, weight=w, bias=b)
    if COUT:
        op.update(post_weight=(np.random.randn(COUT, C) / np.sqrt(C)).astype(np.float32),
                  post_bias=np.zeros(COUT, np.float32), post_act=1, post_cin=C)
    prog.ops.append(op)
prog.ws_bytes = ar.top + 256
eng = HipEngine(prog, dev)
for o in prog.ops:
    eng.write_view(o["inp"], torch.randn(B, C, H, W))
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
byt = B * H * W * (C + (COUT or C)) * 2
print(f"dwconv C={C} -> {COUT or C} @{H}x{W}: {us:7.1f} us  {byt / us / 1e6:6.2f} TB/s")
