#!/bin/bash
# Timing ablations of conv_gemm_kernel's main loop (development): rebuilds conv_gemm.hip with YL_GEMM_ABL = 0 / 2 (no
# operand traffic after the prologue) / 4 (no fragment reads) / 6 on the GPU box and times 768->768 at 160x160 batch 8.
cd "$(dirname "$0")/.."
for abl in ${ABLS:-2 4 6 0}; do
  touch yolo-lite_amd/csrc/conv_gemm.hip
  make -C yolo-lite_amd/csrc EXTRA=-DYL_GEMM_ABL=$abl > /dev/null 2>&1 || exit 1
  for w8 in ${W8S:-0 -1}; do
    echo -n "abl=$abl w8=$w8 "
    YL_MICRO_B=8 YL_CONV_GEMM=1 YL_CONV_GEMM_W8=$w8 YL_CONV_GEMM_BLOCKS=1 python tools/conv_micro.py 768 768 1 1 160 160 2>&1 | tail -1
  done
done
