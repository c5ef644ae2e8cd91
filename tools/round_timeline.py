"""When does each batch of a ONE-ROUND run finish?  (the driver's `bench.py --steps 20` = twenty batches in flight, one round)

usage: python tools/round_timeline.py [--streams 20] [--rounds 1] [--stagger-us 0] [--reps 5]

Builds the bench's engine, prepares one graph per slot exactly as bench.py does and issues `rounds * streams` steps through
yl_graph_launch_many.  An event behind every slot's last graph gives that batch's finish time relative to a start event, so
the output shows whether the batches of a round move in lock step (all finish together) and what the round costs against
streams x the steady-state step.  --stagger-us N: slot j's first launch waits for a spin kernel of j x N microseconds
(torch._sleep), an experiment on de-phasing the round."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=20)
ap.add_argument("--rounds", type=int, default=1)
ap.add_argument("--stagger-us", type=float, default=0.0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--mix", type=int, default=0, help="every MIX-th slot runs the LATENCY profile of the program (0: none): same results, other launches")
args = ap.parse_args()
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(args.streams))
from yololite_amd import _lib  # noqa: E402

_lib.apply_runtime_knobs()
import torch  # noqa: E402

import bench  # noqa: E402  (its module level only sets knobs and constants)

dev = torch.device("cuda:0")
B, H, W = args.batch, 640, 640
x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
specs, sd, eng = bench.build_model("n", dev, B, H, W, x, target_frac=0.015)
S = args.streams
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
xs = [torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(1000 + j), dtype=torch.uint8).to(dev)
      for j in range(S)]
eng_lat = None
if args.mix > 0:
    from yololite_amd.nn.engine import HipEngine
    eng_lat = HipEngine.from_model(specs, sd, B, H, W, dev, one_batch=True)
handles = []
for j in range(S):
    e_j = eng_lat if (eng_lat is not None and j % args.mix == args.mix - 1) else eng
    with torch.cuda.stream(streams[j]):
        _, _, h = e_j.prepare_detect(xs[j], 0.25, 0.7, max_det=300, slot=j)
    handles.append(h)
K = S * args.rounds
step_h = [handles[k % S] for k in range(K)]
step_s = [streams[k % S] for k in range(K)]
eng.launch_many(step_h, step_s)
torch.cuda.synchronize()
eng.launch_many(step_h, step_s)
torch.cuda.synchronize()
res = []
for rep in range(args.reps):
    ev0 = torch.cuda.Event(enable_timing=True)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(streams[0])
    if args.stagger_us > 0:
        for j in range(1, S):
            with torch.cuda.stream(streams[j]):
                torch.cuda._sleep(int(j * args.stagger_us * 1e-6 * 2.4e9))
    eng.launch_many(step_h, step_s)
    t_issue = time.perf_counter() - t0
    for j in range(S):
        evs[j].record(streams[j])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    fin = sorted(ev0.elapsed_time(e) for e in evs)
    res.append({"wall_ms": round(wall * 1e3, 3), "issue_ms": round(t_issue * 1e3, 3), "first_done_ms": round(fin[0], 3),
                "median_done_ms": round(fin[len(fin) // 2], 3), "last_done_ms": round(fin[-1], 3),
                "images_per_s": round(B * K / wall, 1), "finish_ms": [round(f, 2) for f in fin]})
best = min(res, key=lambda r: r["wall_ms"])
line = json.dumps({"mix": args.mix, "streams": S, "rounds": args.rounds, "stagger_us": args.stagger_us, "best": best,
                   "all_images_per_s": [r["images_per_s"] for r in res]})
os.write(bench._REAL_STDOUT, (line + "\n").encode())  # (importing bench sends fd 1 to stderr: its one-JSON-line contract)
