#!/usr/bin/env python3
"""Development measurement (review item 3 ii): two chained 3x3 convs 64 -> 64 at 80x80, batch 32 -- the first two layers of
Detect's box branch on level 0 -- as ONE launch with the intermediate in LDS (bneck2_kernel<8>, whose arithmetic is
x + SiLU(cv2(SiLU(cv1(x)))): the same MFMA / LDS work as the branch plus one add) against the two launches the network uses
(conv3_cs_kernel, conv_kernel).  Usage: python tools/box_branch_micro.py [H W]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.lower import OP_BNECK2, OP_CONV, Program, _Arena

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (80, 80)
B, C, NREP = 32, 64, 6
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
w1 = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
w2 = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
b0 = np.zeros(C, np.float32)


def run(fused: bool) -> float:
    ar = _Arena(B)
    prog = Program(B, 32, 32, 80)
    prog.input_view = ar.alloc(32, 32, 8)
    for r in range(NREP):
        x, t, y = ar.alloc(H, W, C), ar.alloc(H, W, C), ar.alloc(H, W, C)
        if fused:
            prog.ops.append(dict(kind=OP_BNECK2, name=f"b{r}", act=1, inp=x, out=y, res=None, weight=w1, bias=b0, post_weight=w2, post_bias=b0))
        else:
            prog.ops.append(dict(kind=OP_CONV, name=f"a{r}", k=3, stride=1, act=1, inp=x, out=t, res=None, weight=w1, bias=b0))
            prog.ops.append(dict(kind=OP_CONV, name=f"c{r}", k=3, stride=1, act=1, inp=t, out=y, res=None, weight=w2, bias=b0))
    prog.ws_bytes = ar.top + 256
    eng = HipEngine(prog, dev)
    for o in prog.ops:
        if o["name"][0] in "ab":
            eng.write_view(o["inp"], torch.randn(B, C, H, W))
    im = torch.zeros(B, 3, 32, 32, device=dev)
    for _ in range(3):
        eng.forward(im, decode=False, graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.forward(im, decode=False, graph=True)
    e1.record()
    torch.cuda.synchronize()
    kern = sorted({eng.op_kernel(i) for i in range(len(prog.ops))})
    return e0.elapsed_time(e1) / (20 * NREP) * 1e3, kern


for fused in (False, True):
    us, kern = run(fused)
    print(f"{'one launch (bneck2_kernel<8>)' if fused else 'two launches':32s} @{H}x{W} batch {B}: {us:7.1f} us per pair of convs  {kern}")
