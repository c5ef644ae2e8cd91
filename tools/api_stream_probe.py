#!/usr/bin/env python3
"""Development probe: host cost of the pipelined predict(stream=True) path per batch (cProfile of one pass)."""
import cProfile
import os
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
os.environ.setdefault("YL_PREDICT_INFLIGHT", "20")
import torch

from yololite_amd import YOLOLite
from yololite_amd.data.loaders import LoadTensorBatches

dev = torch.device("cuda:0")
S = int(os.environ["YL_PREDICT_INFLIGHT"])
xs = [torch.randint(0, 256, (32, 3, 640, 640), dtype=torch.uint8, generator=torch.Generator().manual_seed(j)).to(dev) for j in range(S)]
m = YOLOLite("yolo11n.yaml")
kw = dict(conf=0.25, iou=0.7, device="cuda:0", save=False, verbose=False, batch=32)
for _ in m.predict(LoadTensorBatches(xs), stream=True, **kw):
    pass
torch.cuda.synchronize()
src = LoadTensorBatches([xs[j % S] for j in range(200)])
import contextlib
side = torch.cuda.Stream()
for label, ctx, nowait in (("default stream", contextlib.nullcontext, "0"), ("no wait_stream", contextlib.nullcontext, "1"),
                           ("side stream current", lambda: torch.cuda.stream(side), "0")):
    os.environ["YL_PREDICT_NOWAIT"] = nowait
    with ctx():
        for rep in range(2):
            t = time.perf_counter()
            n = sum(1 for _ in m.predict(src, stream=True, **kw))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
    print(label, "images/s", round(n / dt, 1), "ms per batch", round(dt / 200 * 1e3, 3))
os.environ["YL_PREDICT_NOWAIT"] = "0"
eng = next(iter(m.predictor.model._engines.values()))
sts = [torch.cuda.Stream() for _ in range(S)]
for rep in range(2):
    t = time.perf_counter()
    for k in range(200):
        with torch.cuda.stream(sts[k % S]):
            eng.detect(xs[k % S], 0.25, 0.7, max_det=300, graph=True, slot=k % S)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
print("engine.detect on", S, "streams: images/s", round(200 * 32 / dt, 1))

# where the host time of one pass goes
pr = cProfile.Profile()
pr.enable()
n = sum(1 for _ in m.predict(src, stream=True, **kw))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
