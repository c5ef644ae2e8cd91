#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 counter_collection CSVs (development aid)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("yl::", "")[:48]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if pat in k:
        print(k, {c: round(v[0] / v[1]) for c, v in sorted(cs.items())}, "launches", max(v[1] for v in cs.values()))
