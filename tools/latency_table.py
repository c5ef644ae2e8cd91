"""Per-op table (in-sequence HIP events) of the LATENCY profile of the program (nn/lower.py lower(one_batch=True)) -- what
YOLOLite.predict() per call runs -- beside its graph replay time.  python tools/latency_table.py [batch]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from yololite_amd import _lib  # noqa: E402

_lib.apply_runtime_knobs()
import bench  # noqa: E402
from yololite_amd.nn.engine import HipEngine  # noqa: E402


def main(B=32, H=640, W=640):
    dev = torch.device("cuda:0")
    x = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(0), dtype=torch.uint8).to(dev)
    specs, sd, eng = bench.build_model("n", dev, B, H, W, x)
    del eng
    torch.cuda.empty_cache()
    eng1 = HipEngine.from_model(specs, sd, B, H, W, dev, one_batch=True)
    rows = eng1.profile_detect(x, 0.25, 0.7, max_det=300, iters=20)
    tot = 0.0
    for r in rows:
        tot += r["ms"]
        print(f"{r['ms'] * 1e3:7.1f} us  {r['kernel']:44s} {r['name']}")
    for _ in range(20):
        eng1.detect(x, 0.25, 0.7, graph=True)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        eng1.detect(x, 0.25, 0.7, graph=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{len(rows)} rows, in-sequence sum {tot * 1e3:.0f} us; graph replay + synchronise {dt * 1e6:.0f} us = {B / dt:.0f} images/s")


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
