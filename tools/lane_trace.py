"""Eager one-batch steps of the latency profile for a rocprofv3 kernel trace (see tools/lane_probe.py): 12 steps, host-synchronised."""
import sys
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
import os
mode = os.environ.get("LANE_TRACE_GRAPH", "0") == "1"
for _ in range(12):
    eng1.detect(x, 0.25, 0.7, graph=mode)
    torch.cuda.synchronize()
