#!/usr/bin/env python3
"""Development probe: K batches issued round-robin on S streams (one engine + workspace per stream)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from yololite_amd.nn.engine import HipEngine  # noqa: E402

dev = torch.device("cuda:0")
x = torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(0)).to(dev)
specs, sd, eng0 = build_model("n", dev, 32, 640, 640, x)
for S in (1, 2, 3):
    engs = [eng0] + [HipEngine.from_model(specs, sd, 32, 640, 640, dev) for _ in range(S - 1)]
    xs = [x.clone() for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]

    def run(k):
        for i in range(k):
            j = i % S
            with torch.cuda.stream(streams[j]):
                engs[j].detect(xs[j], 0.25, 0.7, graph=True)

    run(2 * S)
    torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams {S} (one engine per stream): {dt / K * 1e3:.4f} ms/step  {32 * K / dt:.0f} img/s")

    def run1(k):
        for i in range(k):
            j = i % S
            with torch.cuda.stream(streams[j]):
                eng0.detect(xs[j], 0.25, 0.7, graph=True, slot=j)

    run1(2 * S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run1(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams {S} (one engine, {S} slots)   : {dt / K * 1e3:.4f} ms/step  {32 * K / dt:.0f} img/s")
