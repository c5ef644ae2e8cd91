#!/bin/bash
# Development aid: ablations of conv3_tile_kernel (YL_CONV_DBG bits: 1 no DMA, 2 no stores, 4 no slab copy, 8 no SiLU,
# 16 no MFMA loop, 32 return at once, 64 no epilogue)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cfg="${CFG:-64 64 3 1 80 80}"
for d in ${DBGS:-0 32 95 87 31}; do
  YL_CONV_DBG=$d YL_CONV_TILE=1 YL_CONV_TILE_MIN=100 python3 $R/tools/conv_micro.py $cfg 2>&1 | grep dbg
done
