#!/usr/bin/env python3
"""Development probe: a few graph replays of the hot path, to be run under rocprofv3 --kernel-trace."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from bench import build_model  # noqa: E402

dev = torch.device("cuda:0")
x = torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(0)).to(dev)
specs, sd, eng = build_model("n", dev, 32, 640, 640, x)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng.detect(x, 0.25, 0.7, graph=True)
torch.cuda.synchronize()
