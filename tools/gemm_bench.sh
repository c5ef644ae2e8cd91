#!/bin/bash
# conv_gemm_kernel against conv_kernel on wide 1x1 layers (YOLO11x at 1280x1280 batch 8, YOLO11s at 640x640 batch 64):
# conv_kernel / four-wave tiles (128 pixels) / eight-wave tiles (256 pixels), grid thresholds off.
# Usage: tools/gemm_bench.sh > gpurun_out/gemm_bench.log
cd "$(dirname "$0")/.."
run() {  # B cin cout H W
  for mode in "0 0" "1 0" "1 1"; do
    set -- "$1" "$2" "$3" "$4" "$5" $mode
    echo -n "B=$1 gemm=$6 w8=$7  "
    YL_MICRO_B=$1 YL_CONV_GEMM=$6 YL_CONV_GEMM_W8=$((0-$7)) YL_CONV_GEMM_BLOCKS=1 python tools/conv_micro.py $2 $3 1 1 $4 $5 2>&1 | grep -v amdgpu.ids | tail -1
  done
}
run 8 192 192 320 320
run 8 384 384 320 320
run 8 384 384 160 160
run 8 768 768 160 160
run 8 768 768 80 80
run 8 1536 768 80 80
run 8 768 384 40 40
run 8 1536 768 40 40
run 64 128 128 80 80
run 64 256 256 40 40
run 64 384 256 40 40
run 64 512 512 20 20
run 64 768 512 20 20
