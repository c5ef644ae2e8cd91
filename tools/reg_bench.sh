#!/bin/bash
# Development aid: filter-in-registers 3x3 kernel vs the LDS-tile and the direct kernel (gpurun -- 'bash tools/reg_bench.sh').
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "64 64 3 1 80 80" "64 64 3 1 40 40" "32 32 3 1 40 40" "64 32 3 1 40 40" "32 64 3 1 40 40" "64 64 3 1 20 20"; do
  YL_CONV_REG=0 YL_CONV_TILE=0 python3 $R/tools/conv_micro.py $cfg
  YL_CONV_REG=0 YL_CONV_TILE=1 YL_CONV_TILE_MIN=100 python3 $R/tools/conv_micro.py $cfg
  for nt in 4 2; do
    echo "reg NT=$nt"; YL_CONV_REG=1 YL_CONV_REG_MIN=1 YL_CONV_REG_NT=$nt python3 $R/tools/conv_micro.py $cfg
  done
done
