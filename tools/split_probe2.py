#!/usr/bin/env python3
"""Development probe: one batch-32 engine vs two (four) concurrent engines of batch 16 (8) on separate streams."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from yololite_amd.nn.engine import HipEngine  # noqa: E402

dev = torch.device("cuda:0")
x = torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(0)).to(dev)
specs, sd, eng = build_model("n", dev, 32, 640, 640, x)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("1 x 32:", timeit(lambda: eng.detect(x, 0.25, 0.7, graph=True)))
for parts in (2, 4):
    bs = 32 // parts
    engs = [HipEngine.from_model(specs, sd, bs, 640, 640, dev) for _ in range(parts)]
    xs = [x[i * bs:(i + 1) * bs].contiguous() for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]

    def run():
        for e, xi, s in zip(engs, xs, streams):
            with torch.cuda.stream(s):
                e.detect(xi, 0.25, 0.7, graph=True)

    print(f"{parts} x {bs} concurrent:", timeit(run))
    print(f"{parts} x {bs} serial    :", timeit(lambda: [e.detect(xi, 0.25, 0.7, graph=True) for e, xi in zip(engs, xs)]))
