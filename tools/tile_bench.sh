#!/bin/bash
# Development aid: LDS-tile 3x3 kernel vs the direct kernel on the layers it takes (gpurun -- 'bash tools/tile_bench.sh').
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "64 64 3 1 80 80" "64 64 3 1 40 40" "32 32 3 1 40 40" "64 32 3 1 40 40" "32 64 3 1 40 40"; do
  YL_CONV_TILE=0 python3 $R/tools/conv_micro.py $cfg
  YL_CONV_TILE=1 YL_CONV_TILE_MIN=${TMIN:-100} python3 $R/tools/conv_micro.py $cfg
done
