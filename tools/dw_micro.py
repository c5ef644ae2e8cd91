#!/usr/bin/env python3
"""Stand-alone timing of the depthwise 3x3 conv (ops_misc.hip): usage  python tools/dw_micro.py C H W [batch]
(YL_DW_PX = columns per thread: 1, 2 (default) or 4; YL_MICRO_ACT=0: no SiLU)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch

from yololite_amd import _lib

_lib.apply_runtime_knobs()
from hip_util import Mini, bf16_round  # noqa: E402
from yololite_amd.nn.lower import OP_DWCONV3  # noqa: E402

C, H, W = map(int, sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
NREP = 6
m = Mini(B, dev)
wt = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).numpy()
for r in range(NREP):
    xin, out = m.buf(H, W, C), m.buf(H, W, C)
    if r == 0:
        m.fill(xin, bf16_round(torch.randn(B, C, H, W, generator=g)))
    m.op(kind=OP_DWCONV3, name=f"dw{r}", k=3, stride=1, act=int(os.environ.get("YL_MICRO_ACT", "1")), inp=xin, out=out, weight=wt, bias=np.zeros(C, np.float32))
m.run()
im = torch.zeros(B, 3, 32, 32, device=dev)
for _ in range(3):
    m.eng.forward(im, decode=False, graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 20
for _ in range(N):
    m.eng.forward(im, decode=False, graph=True)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / (N * NREP) * 1e3
byt = 2.0 * B * H * W * C * 2
print(f"dw3x3 {C} ch @{H}x{W} x {B} (YL_DW_PX={os.environ.get('YL_DW_PX', '2')}): {us:7.1f} us  {byt / us / 1e6:5.2f} TB/s")
