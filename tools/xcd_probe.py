#!/usr/bin/env python3
"""Development aid: where do the workgroups of a CU-masked stream land?  (yl_stream_create_masked / yl_xcc_census)"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from yololite_amd import _lib

_lib.apply_runtime_knobs()
import torch

dev = torch.device("cuda:0")


def masked(bits):
    words = (C.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    _lib.check(_lib.lib().yl_stream_create_masked(words, 8, C.byref(out)), "mask")
    return torch.cuda.ExternalStream(out.value, device=dev)


def show(name, st):
    h = _lib.xcc_census(st, 8192)
    per_xcd = h.sum((1, 2)).tolist()
    units = int((h > 0).sum())
    print(f"{name:34s} per XCD {per_xcd}  units used {units}")
    return h


show("default stream", torch.cuda.current_stream(dev))
show("bits 0..31", masked(range(32)))
show("bits 0..7", masked(range(8)))
show("bits i % 8 == 0", masked(range(0, 256, 8)))
show("bits i % 8 in (0, 1)", masked([i for i in range(256) if i % 8 < 2]))
show("bits 0..127", masked(range(128)))
h = show("bits 32..63", masked(range(32, 64)))
print((h > 0).nonzero()[:40].tolist())
