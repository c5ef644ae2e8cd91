#!/usr/bin/env python3
"""Development probe: time of op 0 (the fused stem) alone, for ablation runs (YL_STEM_DBG / YL_STEM_CH)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from bench import build_model  # noqa: E402

dev = torch.device("cuda:0")
dt = {"f32": torch.float32, "u8": torch.uint8, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
x = torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(0))
specs, sd, eng = build_model("n", dev, 32, 640, 640, x.to(dev))
xi = (x * 255).to(torch.uint8) if dt == torch.uint8 else x.to(dt)
rows = eng.profile(xi.to(dev), iters=20)
print(sys.argv[1:], "op0", rows[1]["name"], rows[1]["kernel"], f"{rows[1]['ms'] * 1e3:.1f} us")
