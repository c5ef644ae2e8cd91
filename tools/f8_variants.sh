#!/bin/bash
# conv_f8_kernel variants end to end on YOLO11x e4m3 1280x1280 batch 8 (development): prefetch depth (build switch YL_F8_D,
# rebuilt on the box) x pixel tiles per wave (YL_F8_NT).
cd "$(dirname "$0")/.."
for d in 4 3 2; do
  touch yolo-lite_amd/csrc/conv_f8.hip
  make -C yolo-lite_amd/csrc EXTRA=-DYL_F8_D=$d > /dev/null 2>&1 || exit 1
  for nt in 0 2; do
    echo -n "D=$d NT=$nt: "
    YL_F8_NT=$nt python bench.py --scale x --batch 8 --imgsz 1280 --dtype fp8 --steps 40 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); print(b['value'], b.get('value_sync_per_iteration'), {k:(round(v['ms'],3), v['launches']) for k,v in b['kernels'].items() if 'f8' in k})"
  done
done
