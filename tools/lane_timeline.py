"""Timeline of the LAST step in a rocprofv3 kernel-trace CSV of tools/lane_trace.py: start, duration, queue of every dispatch."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "stem2_kernel" in r["Kernel_Name"])
seq = rows[idx:]
t0 = int(seq[0]["Start_Timestamp"])
endp = t0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void yl::", "").split("(")[0][:48]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {(s - endp) / 1e3:6.1f} q{r['Queue_Id']:>2} {name} grid={r['Grid_Size_X']}")
    endp = max(endp, e)
print("span", (endp - t0) / 1e3)
