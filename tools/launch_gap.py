#!/usr/bin/env python3
"""Development probe: per-kernel floor of dependent tiny kernels, eager stream vs hipGraph replay."""
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
N = 90


def chain():
    for _ in range(N):
        x.add_(1.0)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3 / N, (time.perf_counter() - t0) / n * 1e6 / N


print("eager  us/kernel (gpu, wall):", timeit(chain))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        chain()
print("graph  us/kernel (gpu, wall):", timeit(g.replay))
