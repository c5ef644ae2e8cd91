#!/usr/bin/env python3
"""Per-kernel timeline of the LAST hot-path replay in a rocprofv3 kernel-trace CSV: duration of each
dispatch and the idle gap before it (development aid)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay starts at the last stem kernel
idx = max(i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"] or "convert_input" in r["Kernel_Name"])
seq = rows[idx:]
t0 = int(seq[0]["Start_Timestamp"])
end_prev = t0
busy = 0
print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  queue kernel")
tot_gap = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end_prev) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("yl::", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {r.get('Queue_Id', '?'):>3} {name}  grid={r.get('Grid_Size_X', '?')} wg={r.get('Workgroup_Size_X', '?')}")
    if gap > 0:
        tot_gap += gap
    end_prev = max(end_prev, e)
print(f"span {(end_prev - t0) / 1e3:.1f} us, idle gaps {tot_gap:.1f} us, kernels {len(seq)}")
