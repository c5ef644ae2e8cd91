"""Does a side stream hide the 80x80 Detect branches behind the 40x40 / 20x20 neck when ONE batch runs at a time?
Eager launches (real streams and events; a captured graph's fork edges are serialised by this runtime, yl_graph_branch_probe):
python tools/lane_probe.py  -- run under YL_SIDE_LANES=0/1, YL_LANES=..., YL_GROUP_LEVELS=0; prints the time per batch."""
import os
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

dev = torch.device("cuda:0")
B, H, W = 32, 640, 640
x = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(0), dtype=torch.uint8).to(dev)
specs, sd, eng = bench.build_model("n", dev, B, H, W, x)
del eng
torch.cuda.empty_cache()
eng1 = HipEngine.from_model(specs, sd, B, H, W, dev, one_batch=True)
for mode in (False, True):
    for _ in range(20):
        o, c = eng1.detect(x, 0.25, 0.7, graph=mode)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        o, c = eng1.detect(x, 0.25, 0.7, graph=mode)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    os.write(bench._REAL_STDOUT, (f"lanes={os.environ.get('YL_SIDE_LANES','0')}:{os.environ.get('YL_LANES','')} group={os.environ.get('YL_GROUP_LEVELS','')} "
                                  f"graph={mode} ops={len(eng1.prog.ops)}: {dt * 1e6:.0f} us per batch, {int(c.sum())} detections\n").encode())
