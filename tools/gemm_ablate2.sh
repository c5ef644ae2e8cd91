#!/bin/bash
# conv_gemm_kernel's timing ablations (build switch YL_GEMM_ABL: 2 = no operand traffic, 4 = no fragment reads, 8 = nothing stored;
# wrong results) and the activation's cost (YL_MICRO_ACT=0) on the 1x1 shapes that carry YOLO11x @1280 and YOLO11s @640: rebuilds
# conv_gemm.o per switch ON THE GPU BOX.  Usage: bash tools/gemm_ablate2.sh > gpurun_out/gemm_ablate2.log  (leaves a production build)
cd "$(dirname "$0")/.."
shapes=("8 384 384 320 320" "8 768 768 160 160" "64 128 128 80 80" "64 256 256 40 40")
for abl in ${ABLS:-0 8 2 10 14}; do
  make -C yolo-lite_amd/csrc -B conv_gemm.o EXTRA=-DYL_GEMM_ABL=$abl > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  for sh in "${shapes[@]}"; do
    set -- $sh
    for act in 1 0; do
      echo -n "ABL=$abl act=$act B=$1  "
      YL_MICRO_B=$1 YL_MICRO_ACT=$act YL_CONV_GEMM_BLOCKS=1 python tools/conv_micro.py $2 $3 1 1 $4 $5 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
make -C yolo-lite_amd/csrc -B conv_gemm.o > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1
