"""Per-op table of two bench runs side by side: python tools/perop_diff.py A.json B.json  (files written by bench.py --profile-out)."""
import json
import sys

a, b = (json.load(open(f)) for f in sys.argv[1:3])
bn = {r["name"]: r for r in b}
ta = tb = 0.0
for r in a:
    q = bn.get(r["name"])
    ta += r["ms"]
    if q is None:
        print(f"{r['ms'] * 1e3:8.1f}        -   {r['kernel']:30s} {r['name'][:60]}")
        continue
    tb += q["ms"]
    mark = "  <--" if abs(q["ms"] - r["ms"]) > 0.0015 or q["kernel"] != r["kernel"] else ""
    print(f"{r['ms'] * 1e3:8.1f} {q['ms'] * 1e3:8.1f}   {r['kernel']:30s} {q['kernel']:30s} {r['name'][:50]}{mark}")
for q in b:
    if q["name"] not in {r["name"] for r in a}:
        tb += q["ms"]
        print(f"       - {q['ms'] * 1e3:8.1f}   {'':30s} {q['kernel']:30s} {q['name'][:50]}")
print(f"sum {ta * 1e3:.1f} {tb * 1e3:.1f}")
