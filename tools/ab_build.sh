#!/bin/bash
# A/B of a build switch of one source file across the three bench configurations, ON THE GPU BOX (one box = one device, one clock):
#   bash tools/ab_build.sh conv_gemm "-DYL_GEMM_ST_AUX=0" "-DYL_GEMM_ST_AUX=2"
# rebuilds <file>.o with each EXTRA in turn (twice round: A B A B), runs bench.py on configs 2, 3, 5 and prints value per run.
# The production object is restored on EVERY exit (build failure, timeout, Ctrl-C): an A / B object left in the tree would be
# picked up by the next plain `make` (EXTRA is not a dependency) and the tests / bench would silently run it.
cd "$(dirname "$0")/.."
f=$1; shift
mkdir -p gpurun_out
trap 'make -C yolo-lite_amd/csrc -B $f.o > /dev/null 2>&1; make -C yolo-lite_amd/csrc > /dev/null 2>&1' EXIT
for rep in 1 2; do
  for extra in "$@"; do
    make -C yolo-lite_amd/csrc -B $f.o EXTRA="$extra" > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1 || { echo "build failed: $extra"; exit 1; }
    for cfg in "" "--scale s --batch 64" "--scale x --imgsz 1280 --batch 8 --dtype fp8"; do
      python bench.py $cfg --no-cpu-baseline --no-other-configs > gpurun_out/ab.json 2> gpurun_out/ab.err
      python - "$extra" "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {sys.argv[2] or 'n 640 b32':44s} value {d['value']:9.1f}  sync/iter {d.get('value_sync_per_iteration', 0):9.1f}", flush=True)
PY
    done
  done
done
