#!/bin/bash
# Registers, spills and scratch of every kernel of one source file (hipcc -S of the device code, no GPU needed):
#   [TOP=12] bash tools/vgpr_report.sh conv_mfma [extra flags]   -> name, vgpr, spilled vgprs, scratch bytes; sorted by spills then vgprs
f=$1; shift
cd "$(dirname "$0")/../yolo-lite_amd/csrc"
unroll=""
case $f in conv3_reg|conv3_pair|conv3_cs|conv3_sw) unroll="-mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000";; esac
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off $unroll "$@" --cuda-device-only -S -o /tmp/vgpr_$f.s $f.hip || exit 1
python3 - /tmp/vgpr_$f.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
rows = []
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    body = m.group(2)
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, body) else 0
    rows.append((g("vgpr_spill_count"), g("vgpr_count"), g("private_segment_fixed_size"), m.group(1)))
rows.sort(reverse=True)
print(f"{len(rows)} kernels; spilling: {sum(1 for r in rows if r[0] or r[2])}; mean vgpr {sum(r[1] for r in rows) / max(1, len(rows)):.1f}")
for sp, v, sc, n in rows[:int(__import__("os").environ.get("TOP", "12"))]:
    print(f"  vgpr {v:3d} spill {sp:3d} scratch {sc:5d}  {n[:110]}")
PY
