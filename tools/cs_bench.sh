#!/bin/bash
# Development aid: the Cout-split filter-in-registers kernel (conv3_cs) against conv3_reg and conv_kernel, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "64 64 3 1 80 80" "64 64 3 1 40 40"; do
  for rep in 1 2; do
    echo -n "direct "; YL_CONV_CS=0 YL_CONV_REG=0 python3 $R/tools/conv_micro.py $cfg 2>&1 | tail -1
    echo -n "reg    "; YL_CONV_CS=0 YL_CONV_REG=1 YL_CONV_REG_MIN=1 python3 $R/tools/conv_micro.py $cfg 2>&1 | tail -1
    echo -n "cs     "; YL_CONV_CS=1 YL_CONV_CS_MIN=1 python3 $R/tools/conv_micro.py $cfg 2>&1 | tail -1
  done
done
