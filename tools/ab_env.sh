#!/bin/bash
# A/B of environment switches across the three bench configurations on ONE box (one device, one clock), two rounds:
#   bash tools/ab_env.sh "YL_ST_NT=0" "YL_ST_NT=-1" "YL_ST_NT=1"        (CFGS="2 3 5" selects the configurations)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for envs in "$@"; do
    for c in ${CFGS:-2 3 5}; do
      case $c in 2) cfg="";; 3) cfg="--scale s --batch 64";; 5) cfg="--scale x --imgsz 1280 --batch 8 --dtype fp8";; esac
      env $envs python bench.py $cfg --no-cpu-baseline --no-other-configs > gpurun_out/ab.json 2> gpurun_out/ab.err
      python - "$envs" "config $c" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {sys.argv[2]:10s} value {d['value']:9.1f}  sync/iter {d.get('value_sync_per_iteration', 0):9.1f}", flush=True)
PY
    done
  done
done
