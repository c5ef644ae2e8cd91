"""HBM traffic PER LAYER from rocprofv3 counter collection: the dispatches of one step (one hardware queue, stem launch to stem launch)
are matched by position with the per-op table of the same bench configuration.

  on the GPU box (default stream count: `--streams 1` under counter collection crashed the profiled process in round 4):
    python3 bench.py <config> --profile-out DIR/per_op.json > DIR/bench.json
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR/pmc_fetch -- python3 bench.py <config> --steps 20 --warmup 1 --no-cpu-baseline --no-other-configs
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d DIR/pmc_write -- python3 bench.py <config> ...
  here:  python tools/pmc_per_layer.py DIR [substring of the kernel names to list, default f8]

Read side = 2 x FETCH_SIZE (gfx950, MI355X_MICROARCH.md); algorithmic bytes = the engine's per-op figure (`yl_engine_op_info`).
"""
import collections
import csv
import glob
import json
import sys

import numpy as np


def load(path, counter):
    per_q = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                per_q[row["Queue_Id"]].append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    return {q: sorted(v) for q, v in per_q.items()}


def steps_of(per_q):
    out = []
    for lst in per_q.values():
        idx = [i for i, (_, k, _) in enumerate(lst) if "yl::stem" in k]
        out += [lst[a:b] for a, b in zip(idx, idx[1:])]
    n = collections.Counter(len(s) for s in out).most_common(1)[0][0]
    return [s for s in out if len(s) == n]


def main(d, pat="f8"):
    fe = steps_of(load(glob.glob(f"{d}/pmc_fetch/*/*_counter_collection.csv")[0], "FETCH_SIZE"))
    wr = steps_of(load(glob.glob(f"{d}/pmc_write/*/*_counter_collection.csv")[0], "WRITE_SIZE"))
    rows = json.load(open(f"{d}/per_op.json"))
    rows = rows["rows"] if isinstance(rows, dict) else rows
    F = np.array([[v for _, _, v in s] for s in fe]).mean(0) * 1024 * 2
    W = np.array([[v for _, _, v in s] for s in wr]).mean(0) * 1024
    kn = [k.replace("void yl::", "").split("(")[0] for _, k, _ in fe[0]]
    print(f"{len(fe)} steps of {len(kn)} dispatches; {len(rows)} per-op rows")
    print(f"{'op':34s} {'kernel':40s} {'alg MB':>8s} {'read MB':>8s} {'write MB':>8s} {'pmc/alg':>7s} {'us':>7s}")
    ta = tp = 0.0
    for i, r in enumerate(rows):
        if i >= len(kn) or pat not in kn[i]:
            continue
        a, p = r["bytes"] / 1e6, (F[i] + W[i]) / 1e6
        ta, tp = ta + a, tp + p
        print(f"{r['name'][:34]:34s} {kn[i][:40]:40s} {a:8.1f} {F[i] / 1e6:8.1f} {W[i] / 1e6:8.1f} {p / max(a, 1e-9):7.2f} {r['ms'] * 1e3:7.1f}")
    print(f"total: algorithmic {ta:.0f} MB, PMC {tp:.0f} MB = {tp / ta:.2f} x")


if __name__ == "__main__":
    main(*sys.argv[1:3])
