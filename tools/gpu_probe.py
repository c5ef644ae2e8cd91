#!/usr/bin/env python3
"""Development probe: times the pieces of the hot path separately on the GPU (not part of the bench contract)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from yololite_amd.utils import ops as hops  # noqa: E402


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", default="n")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, H = a.batch, a.imgsz
    x = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(0)).to(dev)
    specs, sd, eng = build_model(a.scale, dev, B, H, H, x)
    y = eng.forward(x, graph=True)
    print("forward graph   (gpu ms, wall ms):", timeit(lambda: eng.forward(x, graph=True)))
    print("forward eager   (gpu ms, wall ms):", timeit(lambda: eng.forward(x, graph=False)))
    print("forward nodecode(gpu ms, wall ms):", timeit(lambda: eng.forward(x, graph=False, decode=False)))
    print("nms             (gpu ms, wall ms):", timeit(lambda: hops.nms_padded(y, 0.25, 0.7)))
    out, cnt = hops.nms_padded(y, 0.25, 0.7)
    cands = (y[:, 4:].amax(1) > 0.25).sum(1).float()
    print("candidates/img mean", cands.mean().item(), "max", cands.max().item(), "dets/img", cnt.float().mean().item())
    print("e2e unfused     (gpu ms, wall ms):", timeit(lambda: hops.nms_padded(eng.forward(x, graph=True), 0.25, 0.7)))
    print("e2e fused detect(gpu ms, wall ms):", timeit(lambda: eng.detect(x, 0.25, 0.7, graph=True)))
    rows = eng.profile(x, iters=20)
    tot = sum(r["ms"] for r in rows)
    print(f"sum of per-op times {tot:.3f} ms over {len(rows)} launches")
    agg = {}
    for r in rows:
        a_ = agg.setdefault(r["kernel"], [0.0, 0])
        a_[0] += r["ms"]
        a_[1] += 1
    print("by kernel:", ", ".join(f"{k} {v[0] * 1e3:.0f}us x{v[1]}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])))
    excess_table(rows, a.top)
    rows.sort(key=lambda r: -r["ms"])
    for r in rows[: (a.top if os.environ.get('YL_PROBE_LIST') else 0)]:
        print(f"{r['name']:26s} {r['kernel']:16s} {r['ms'] * 1e3:8.1f} us {r['flops'] / max(r['ms'], 1e-9) / 1e9:7.1f} TF/s "
              f"{r['bytes'] / max(r['ms'], 1e-9) / 1e6:8.1f} GB/s")




def excess_table(rows, top=40):
    """ops ranked by time above an optimistic roofline (5 TB/s, 1.25 PFLOP/s)."""
    out = []
    for r in rows:
        ideal = max(r["bytes"] / 5e12, r["flops"] / 1.25e15) * 1e3
        out.append((r["ms"] - ideal, r["ms"], ideal, r["name"], r["kernel"]))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    print(f"total excess over roofline: {tot * 1e3:.0f} us")
    for ex, ms, ideal, name, k in out[:top]:
        print(f"  {name:26s} {k:16s} {ms * 1e3:7.1f} us  ideal {ideal * 1e3:6.1f}  excess {ex * 1e3:6.1f}")


if __name__ == "__main__":
    main()
