#!/usr/bin/env python3
"""Per-kernel and whole-step resource sums from the rocprofv3 --pmc passes of tools/pmc_step.sh.

Counter units (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE = cycles summed over the 8 XCDs; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES
/ SQ_WAIT_* = quad-cycles summed over all SIMDs; SQ_VALU_MFMA_BUSY_CYCLES = cycles summed over all SIMDs.  A resource's
"busy" figure below is its busy time averaged over the chip's 1024 SIMDs, in microseconds at the kernel's own effective
clock -- what the resource would need if every kernel of other batches could overlap with it perfectly."""
import csv
import glob
import json
import sys
from collections import defaultdict

NSIMD = 1024
acc = defaultdict(lambda: defaultdict(float))  # kernel -> counter -> sum over dispatches
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("yl::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
# dispatches per step: the bench ran warmup 1 + 3 steps + calibration etc.; normalise every kernel by its own dispatch count
rows = []
for k, cs in acc.items():
    n = max(cnt[k].values())
    g = lambda c: cs.get(c, 0.0) / max(1, cnt[k].get(c, 1))
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    if cyc <= 0:
        continue
    rows.append(dict(kernel=k[:60], dispatches=n, cycles=cyc,
                     valu=g("SQ_ACTIVE_INST_VALU") * 4 / NSIMD, mfma=g("SQ_VALU_MFMA_BUSY_CYCLES") / NSIMD,
                     lds=g("SQ_ACTIVE_INST_LDS") * 4 / NSIMD, vmem=g("SQ_ACTIVE_INST_VMEM") * 4 / NSIMD,
                     waves=g("SQ_WAVE_CYCLES") * 4 / NSIMD, wait=g("SQ_WAIT_ANY") * 4 / NSIMD,
                     wait_inst=g("SQ_WAIT_INST_ANY") * 4 / NSIMD, active=g("SQ_ACTIVE_INST_ANY") * 4 / NSIMD,
                     i_valu=g("SQ_INSTS_VALU"), i_trans=g("SQ_INSTS_VALU_TRANS"), i_mfma=g("SQ_INSTS_MFMA"),
                     i_lds=g("SQ_INSTS_LDS"), i_vrd=g("SQ_INSTS_VMEM_RD"), i_vwr=g("SQ_INSTS_VMEM_WR"), i_salu=g("SQ_INSTS_SALU")))
rows.sort(key=lambda r: -r["cycles"] * r["dispatches"])
print(f"{'kernel':60s} {'n':>5s} {'kcyc':>8s} {'VALU%':>6s} {'MFMA%':>6s} {'LDS%':>6s} {'VMEM%':>6s} {'occ':>5s} {'trans/valu':>10s}")
for r in rows:
    c = r["cycles"]
    print(f"{r['kernel']:60s} {r['dispatches']:5d} {c / 1e3:8.1f} {100 * r['valu'] / c:6.1f} {100 * r['mfma'] / c:6.1f} "
          f"{100 * r['lds'] / c:6.1f} {100 * r['vmem'] / c:6.1f} {r['waves'] / c:5.2f} {r['i_trans'] / max(1.0, r['i_valu']):10.3f}")
# whole run: every dispatch of every kernel (the bench's step mix), resource busy time over serialised kernel time
tot = defaultdict(float)
for r in rows:
    for key in ("cycles", "valu", "mfma", "lds", "vmem", "waves", "wait", "wait_inst", "active"):
        tot[key] += r[key] * r["dispatches"]
c = tot["cycles"]
print()
print(f"all dispatches: serialised kernel cycles {c / 1e6:.2f} M;  busy share of that, chip average:  VALU {100 * tot['valu'] / c:.1f} %  "
      f"MFMA {100 * tot['mfma'] / c:.1f} %  LDS issue {100 * tot['lds'] / c:.1f} %  VMEM issue {100 * tot['vmem'] / c:.1f} %  "
      f"waves per SIMD {tot['waves'] / c:.2f}  (wave time: parked {100 * tot['wait'] / max(1.0, tot['waves']):.0f} %, "
      f"issue-stalled {100 * tot['wait_inst'] / max(1.0, tot['waves']):.0f} %, issuing {100 * tot['active'] / max(1.0, tot['waves']):.0f} %)")
