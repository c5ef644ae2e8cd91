#!/bin/bash
# conv_gemm_kernel on 3x3 convs against conv_kernel (YL_CONV_GEMM3=0/1): YOLO11s's stride-2 layers at batch 64, then end to end.
cd "$(dirname "$0")/.."
for shape in "64 128 2 160 160" "128 256 2 80 80" "256 512 2 40 40" "128 128 2 80 80" "128 128 1 40 40"; do
  set -- $shape
  for g3 in 0 1; do
    echo -n "gemm3=$g3 "
    YL_MICRO_B=64 YL_CONV_GEMM3=$g3 python tools/conv_micro.py $1 $2 3 $3 $4 $5 2>&1 | tail -1
  done
done
bash tools/ab_env.sh YL_CONV_GEMM3 0 1 2 --scale s --batch 64 --steps 200
bash tools/ab_env.sh YL_CONV_GEMM3 0 1 1 --steps 500
