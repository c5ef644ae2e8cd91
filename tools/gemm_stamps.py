#!/usr/bin/env python3
"""Where conv_gemm_kernel's waves wait: clocks in the counted wait for their own LDS-DMA pieces and in the barrier behind it, against
the wave's lifetime.  DIAGNOSTIC build only:

    make -C yolo-lite_amd/csrc -B conv_gemm.o EXTRA=-DYL_STAMPS && make -C yolo-lite_amd/csrc && python tools/gemm_stamps.py cin cout H W [batch]
"""
import ctypes as C
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
from yololite_amd.nn.lower import OP_CONV  # noqa: E402

cin, cout, H, W = map(int, sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
NWG, NWV = 1024, 8
buf = torch.zeros(NWG * NWV * 8, dtype=torch.int64, device=dev)
_lib.lib().yl_diag_set_stamps(C.c_void_p(buf.data_ptr()))
os.environ["YL_CONV_GEMM_BLOCKS"] = "1"
m = Mini(B, dev)
NREP = 4
for r in range(NREP):
    xin, y = m.buf(H, W, cin), m.buf(H, W, cout)
    if r == 0:
        m.fill(xin, bf16_round(torch.randn(B, cin, H, W, generator=gen)))
    m.op(kind=OP_CONV, name=f"c{r}", k=1, stride=1, act=1, inp=xin, out=y, group=0, weight=np.ascontiguousarray((torch.randn(cout, cin, 1, 1, generator=gen) / cin ** 0.5).numpy()),
         bias=np.zeros(cout, np.float32))
m.run()
print("kernel:", m.eng.op_kernel(0))
im = torch.zeros(B, 3, 32, 32, device=dev)
for _ in range(3):
    m.eng.forward(im, decode=False)
torch.cuda.synchronize()
st = buf.view(NWG, NWV, 8).cpu().numpy().astype(np.int64)
live = st[:, :, 3] > 0
if not live.any():
    sys.exit("no stamps: conv_gemm.o is not a diagnostic object")
w, b, n, life, rounds = (st[:, :, i][live].astype(np.float64) for i in range(5))
print(f"{int(live.sum())} waves; tiles per workgroup {np.median(rounds):.0f}; K-slices entered per wave {np.median(n):.0f}")
print(f"lifetime            median {np.median(life):9.0f} clocks")
print(f"counted wait (own pieces) median {np.median(w):9.0f} = {100 * np.median(w / life):4.1f} % of the lifetime  (per slice {np.median(w / n):6.0f})")
print(f"barrier behind it         median {np.median(b):9.0f} = {100 * np.median(b / life):4.1f} %                  (per slice {np.median(b / n):6.0f})")
print(f"the rest (k-loop, epilogue, prologue): {100 * np.median((life - w - b) / life):4.1f} %")
