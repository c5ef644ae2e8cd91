#!/bin/bash
# End-to-end A/B of conv_gemm_kernel (YL_CONV_GEMM=0/1), alternating runs on one box: BASELINE.json configs 5, 3 and 2.
cd "$(dirname "$0")/.."
one() {  # label, env, args...
  local label=$1 gm=$2; shift 2
  YL_CONV_GEMM=$gm python bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label gemm=$gm', d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'sync', d.get('value_sync_per_iteration'), {k:(v['ms'],v['launches']) for k,v in d['kernels'].items() if 'gemm' in k or k=='conv1x1_mfma'})
"
}
for rep in 1 2; do
  for gm in 0 1; do
    one x1280fp8 $gm --scale x --batch 8 --imgsz 1280 --dtype fp8 --steps 60
  done
done
for rep in 1 2; do
  for gm in 0 1; do
    one s640 $gm --scale s --batch 64 --steps 200
  done
done
for rep in 1 2; do
  for gm in 0 1; do
    one n640 $gm --steps 500
  done
done
