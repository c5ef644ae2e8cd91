#!/usr/bin/env python3
"""Development aid: host cost of one hot-path launch (hipGraphLaunch through the C ABI) against the step time."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
B, H, W, S = 32, 640, 640, 5
x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
specs, sd, eng = bench.build_model("n", dev, B, H, W, x)
xs = [torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8).to(dev) for _ in range(S)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
for k in range(20):
    with torch.cuda.stream(streams[k % S]):
        eng.detect(xs[k % S], 0.25, 0.7, graph=True, slot=k % S)
torch.cuda.synchronize()
N = 1000
t0, c0 = time.perf_counter(), time.process_time()
for k in range(N):
    with torch.cuda.stream(streams[k % S]):
        eng.detect(xs[k % S], 0.25, 0.7, graph=True, slot=k % S)
t1, c1 = time.perf_counter(), time.process_time()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue loop {1e3 * (t1 - t0) / N:.3f} ms/step wall, {1e3 * (c1 - c0) / N:.3f} ms/step cpu; to completion {1e3 * (t2 - t0) / N:.3f} ms/step")
# launch cost alone: one stream, GPU idle enough not to back-pressure
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(50):
    eng.detect(xs[0], 0.25, 0.7, graph=True, slot=0)
    torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"launch + sync, one at a time: {1e3 * (t1 - t0) / 50:.3f} ms")
# five launches from an idle GPU (no back-pressure): pure host cost
import statistics
ts = []
for rep in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            eng.detect(xs[k], 0.25, 0.7, graph=True, slot=k)
    t1 = time.perf_counter()
    ts.append((t1 - t0) / S)
print(f"host cost per launch from idle: median {1e3 * statistics.median(ts):.3f} ms, min {1e3 * min(ts):.3f} ms")
