#!/usr/bin/env python3
"""Development probe: host cost of YOLOLite.predict (one batch per call) -- cProfile of 200 calls on a resident uint8 batch."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from yololite_amd import YOLOLite
from yololite_amd.nn.synth import synth_state_dict

dev = torch.device("cuda:0")
m = YOLOLite("yolo11n.yaml")
x = torch.randint(0, 256, (32, 3, 640, 640), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
if "--dense" in sys.argv:  # a saturated head: thousands of candidates per image, 300 detections each (the NMS's worst case)
    m.model.load_state_dict(synth_state_dict("n", 0, cls_bias=[-6.0, -6.0, -6.0], cls_gain=8.0))
else:  # bench.py's head: ~126 candidates per image
    import bench

    _, sd, eng = bench.build_model("n", dev, 32, 640, 640, x)
    del eng
    m.model.load_state_dict(sd)
kw = dict(conf=0.25, iou=0.7, device="cuda:0", save=False, verbose=False, batch=32)
for _ in range(5):
    r = m.predict(x, **kw)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    r = m.predict(x, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("predict: images/s", round(200 * 32 / dt, 1), "ms per call", round(dt / 200 * 1e3, 3), "detections", sum(len(q.boxes) for q in r))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    r = m.predict(x, **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
