#!/usr/bin/env python3
"""Development aid: per-workgroup phase timeline of dwpw_kernel (YL_DWPW_DBG=32 stamps, 10 ns units).
Usage: YL_DWPW_DBG=32 python tools/dwpw_timeline.py C H W Cout"""
import ctypes as C
import os
import sys
from pathlib import Path

os.environ.setdefault("YL_DWPW_DBG", "32")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from yololite_amd import _lib
from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.lower import OP_DWCONV3, Program, _Arena

Cc, H, W, COUT = map(int, sys.argv[1:5])
B = 32
dev = torch.device("cuda:0")
ar = _Arena(B)
prog = Program(B, 32, 32, 80)
prog.input_view = ar.alloc(32, 32, 8)
w = (np.random.randn(Cc, 1, 3, 3) / 3).astype(np.float32)
x, y = ar.alloc(H, W, Cc), ar.alloc(H, W, COUT)
prog.ops.append(dict(kind=OP_DWCONV3, name="d", act=1, inp=x, out=y, res=None, weight=w, bias=np.zeros(Cc, np.float32),
                     post_weight=(np.random.randn(COUT, Cc) / np.sqrt(Cc)).astype(np.float32),
                     post_bias=np.zeros(COUT, np.float32), post_act=1, post_cin=Cc))
prog.ws_bytes = ar.top + 256
eng = HipEngine(prog, dev)
eng.write_view(x, torch.randn(B, Cc, H, W))
im = torch.zeros(B, 3, 32, 32, device=dev)
for _ in range(3):
    eng.forward(im, decode=False, graph=False)
torch.cuda.synchronize()
nwg = min(4096, (B * H * W + 127) // 64)
buf = (C.c_ulonglong * (4096 * 8))()
L = _lib.lib()
L.yl_debug_timestamps.restype = C.c_int
L.yl_debug_timestamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert L.yl_debug_timestamps(buf, 4096 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
t = (t - t0) * 10  # ns
t[t < 0] = -1
print(f"{len(t)} workgroups; kernel span {(t[:, :7].max()) / 1000:.2f} us")
names = ["start", "slab+barrier", "dw done", "barrier", "gemm done", "stores issued", "stores done"]
for k, nm in enumerate(names):
    v = t[:, k]
    print(f"{nm:14s} p5 {np.percentile(v, 5) / 1000:7.2f}  p50 {np.percentile(v, 50) / 1000:7.2f}  p95 {np.percentile(v, 95) / 1000:7.2f}  max {v.max() / 1000:7.2f} us")
d = np.diff(t[:, :7], axis=1)
for k, nm in enumerate(["slab copy", "dw phase", "barrier wait", "gemm", "epilogue issue", "store drain"]):
    print(f"  {nm:14s} mean {d[:, k].mean() / 1000:6.2f}  p95 {np.percentile(d[:, k], 95) / 1000:6.2f} us")
late = (t[:, 0] > 2000).sum()
print(f"workgroups starting later than 2 us after the first: {late}")
