#!/usr/bin/env python3
"""Reduce a rocprofv3 output directory ON THE GPU BOX to what tools/summarize_profile.py needs, then delete the raw CSVs
(gpurun copies back 64 MiB at most; a counter pass of bench.py writes one row per dispatch and counter: 100-200 MB).

    python3 tools/reduce_pmc.py DIR            # DIR = the -d directory of one rocprofv3 run

Writes DIR/agg.json:
  per_kernel: {kernel symbol (short): {counter: [sum, launches]}}                       -- every dispatch of the run
  steps:      {counter: {"n_steps", "kernels": [...], "mean": [...]}}                    -- per launch POSITION of a step (one hardware
              queue, stem launch to stem launch, segments of the modal length): the per-launch table's PMC columns
  trace:      the same per position for the kernel-trace durations (us), when DIR holds a kernel trace
"""
import csv
import glob
import json
import sys
from collections import Counter, defaultdict
from pathlib import Path


def short(name):
    return name.split("(")[0].replace("void ", "").replace("yl::", "")[:60]


def steps(per_queue):
    segs = []
    for lst in per_queue.values():
        # (runtime copy kernels and the NMS counter-clearing launch of the eager / API paths are not launches of the program)
        lst[:] = [r for r in lst if "zero_i32_kernel" not in r[1] and "__amd_rocclr" not in r[1]]
        lst.sort()
        idx = [i for i, r in enumerate(lst) if "stem" in r[1]]
        segs += [lst[a:b] for a, b in zip(idx, idx[1:])]
    if not segs:
        return []
    n = Counter(len(x) for x in segs).most_common(1)[0][0]
    return [x for x in segs if len(x) == n]


def position_means(per_queue):
    st = steps(per_queue)
    if not st:
        return None
    n = len(st[0])
    mean = [sum(s[i][2] for s in st) / len(st) for i in range(n)]
    return {"n_steps": len(st), "kernels": [short(r[1]) for r in st[0]], "mean": mean}


def main(d):
    d = Path(d)
    out = {"per_kernel": {}, "steps": {}}
    for f in glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True):
        per_kernel = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        per_q = defaultdict(lambda: defaultdict(list))
        with open(f) as fi:
            for r in csv.DictReader(fi):
                c, k, v = r["Counter_Name"], r["Kernel_Name"], float(r["Counter_Value"])
                a = per_kernel[short(k)][c]
                a[0] += v
                a[1] += 1
                per_q[c][r["Queue_Id"]].append((int(r["Dispatch_Id"]), k, v))
        for k, cs in per_kernel.items():
            out["per_kernel"].setdefault(k, {}).update({c: v for c, v in cs.items()})
        for c, q in per_q.items():
            pm = position_means(q)
            if pm:
                out["steps"][c] = pm
        Path(f).unlink()
    for f in glob.glob(str(d / "**" / "*kernel_trace.csv"), recursive=True):
        q = defaultdict(list)
        with open(f) as fi:
            for r in csv.DictReader(fi):
                q[r.get("Queue_Id", "0")].append((int(r["Dispatch_Id"]), r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        pm = position_means(q)
        if pm:
            out["trace"] = pm
        Path(f).unlink()
    for f in glob.glob(str(d / "**" / "*.db"), recursive=True):
        Path(f).unlink()
    (d / "agg.json").write_text(json.dumps(out))
    print(f"{d}: {len(out['per_kernel'])} kernels, step tables for {sorted(out['steps'])}{' + trace' if 'trace' in out else ''}")


if __name__ == "__main__":
    main(sys.argv[1])
