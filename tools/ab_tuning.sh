#!/bin/bash
# A/B of DEVELOPMENT switches (tune_int, csrc/yl_internal.h) on the GPU box: builds a -DYL_TUNING copy of the library into
# /tmp (the production .so in the tree is not touched: YL_AMD_LIB selects the copy) and runs bench.py under each environment.
#   bash tools/ab_tuning.sh "--scale s --batch 64 --steps 200" "YL_CONV_KORD_S1=0" "YL_CONV_KORD_S1=128"
# Two rounds (A B A B); prints value and the sync-per-iteration figure per run.
cd "$(dirname "$0")/.."
BA=$1; shift
T=/tmp/yl_tuning
# (the Makefile writes ../libyl_amd.so and includes ../../include: the same relative layout under $T)
rm -rf $T && mkdir -p $T/yl && cp -r yolo-lite_amd/csrc $T/yl/csrc && cp -r include $T/include
( cd $T/yl/csrc && rm -f *.o && make -j16 EXTRA=-DYL_TUNING > $T/build.log 2>&1 ) || { tail -5 $T/build.log; echo "tuning build failed"; exit 1; }
ls -la $T/yl/libyl_amd.so || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
  for envs in "$@"; do
    env YL_AMD_LIB=$T/yl/libyl_amd.so $envs python bench.py $BA --no-cpu-baseline --no-other-configs > gpurun_out/abt.json 2> gpurun_out/abt.err || { tail -3 gpurun_out/abt.err; continue; }
    python - "$envs" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abt.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} value {d['value']:9.1f}  sync/iter {d.get('value_sync_per_iteration', 0):9.1f}", flush=True)
PY
  done
done
