#!/usr/bin/env python3
"""Stand-alone timing of Detect's first box convs on the small levels: the pair 128 -> 64 @40x40 and 256 -> 64 @20x20, batch 32,
on conv3_band_kernel (YL_CONV_BAND=1) and on conv_group_kernel (=0).  Usage: python tools/band_micro.py [batch]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
shapes = [(128, 40, 40), (256, 20, 20)]
import ctypes as C
stamp_buf = torch.zeros(256 * 8 * 32 * 2, dtype=torch.int64, device=dev)
stamps_on = _lib.lib().yl_diag_set_stamps(C.c_void_p(stamp_buf.data_ptr()))
for mode in ("1", "0"):
    os.environ["YL_CONV_BAND"] = mode
    m = Mini(B, dev)
    for c, h, w in shapes:
        xin, y = m.buf(h, w, c), m.buf(h, w, 64)
        m.fill(xin, bf16_round(torch.randn(B, c, h, w, generator=gen)))
        wt = torch.randn(64, c, 3, 3, generator=gen) / (c * 9) ** 0.5
        m.op(kind=OP_CONV, name=f"c{c}", k=3, stride=1, act=1, inp=xin, out=y, group=1, weight=np.ascontiguousarray(wt.numpy()),
             bias=np.zeros(64, np.float32))
    m.run()
    im = torch.zeros(B, 3, 32, 32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(20):
        ev[0].record()
        m.eng.forward(im, decode=False)
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    print(m.eng.op_kernel(0), f"median {sorted(ts)[len(ts) // 2]:.1f} us  min {min(ts):.1f} us (event to event, includes ~6 us of launch overhead)")
    if mode == "1" and stamps_on:
        nwg = B * 5 + B * 2
        st = stamp_buf.view(256, 8, 32, 2)[:nwg].cpu().numpy().astype(np.int64)
        t = (st[..., 1] - st[:, :, :1, 1].min(axis=1, keepdims=True)) * 10 / 1e3  # us since the workgroup's first wave started
        t0 = st[:, :, 0, 1].min()
        for tag, sl in (("128->64 @40x40 (5 bands per image)", slice(0, B * 5)), ("256->64 @20x20 (2 bands per image)", slice(B * 5, nwg))):
            print(tag, "workgroup start spread", (st[sl, 0, 0, 1].max() - t0) * 10 / 1e3, "us")
            for i, nm in enumerate(["start", "staged", "barrier", "k-loop done", "stores issued"]):
                print(f"   {nm:14s} median {np.median(t[sl, :, i]):6.2f} us  max {t[sl, :, i].max():6.2f} us")
