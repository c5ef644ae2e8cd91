#!/usr/bin/env python3
"""Experiment: run the batch as S concurrent slices on S streams (does overlapping the latency-bound
small layers of different slices pay?)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import build_model
from yololite_amd.nn.engine import HipEngine

dev = torch.device("cuda:0")
B, H = 32, 640
x = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(0)).to(dev)
specs, sd, eng = build_model("n", dev, B, H, H, x)


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


print("1 x 32 graph:", timeit(lambda: eng.detect(x, graph=True)))
for S in (2, 4):
    b = B // S
    engs = [HipEngine.from_model(specs, sd, b, H, H, dev) for _ in range(S)]
    xs = [x[i * b:(i + 1) * b].contiguous() for i in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]

    def run():
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        for e, xi, st in zip(engs, xs, streams):
            with torch.cuda.stream(st):
                e.forward(xi, graph=True, decode=False)
        for st in streams:
            cur.wait_stream(st)

    print(f"{S} x {b} concurrent forward(nodecode):", timeit(run))
print("1 x 32 forward(nodecode):", timeit(lambda: eng.forward(x, graph=True, decode=False)))
