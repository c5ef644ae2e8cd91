#!/bin/bash
# conv_gemm_kernel, one tile per workgroup against persistent workgroups (YL_CONV_GEMM_PERSIST), four- and eight-wave tiles.
cd "$(dirname "$0")/.."
for shape in "8 768 768 160 160" "8 384 384 320 320" "8 1536 768 80 80" "64 256 256 40 40" "64 512 512 20 20"; do
  set -- $shape
  for mode in "0 0 0" "1 0 0" "1 -1 0"; do
    set -- $1 $2 $3 $4 $5 $mode
    echo -n "B=$1 persist=$6 w8=$7 dbg=$8  "
    YL_CONV_DBG=$8 YL_MICRO_B=$1 YL_CONV_GEMM=1 YL_CONV_GEMM_PERSIST=$6 YL_CONV_GEMM_W8=$7 YL_CONV_GEMM_BLOCKS=1 python tools/conv_micro.py $2 $3 1 1 $4 $5 2>&1 | tail -1
  done
done
