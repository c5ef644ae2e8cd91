#!/bin/bash
# Ring shapes of conv_gemm_kernel's 128-channel blocks (build switches YL_GM4_KS / _NS; four waves, even K) on YOLO11s batch 64's
# 1x1 shapes: rebuilds conv_gemm.o ON THE GPU BOX and leaves a production build.
cd "$(dirname "$0")/.."
for cfg in "2 2" "1 4" "1 3" "2 2" "1 4"; do
  set -- $cfg
  make -C yolo-lite_amd/csrc -B conv_gemm.o EXTRA="-DYL_GM4_KS=$1 -DYL_GM4_NS=$2" > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  for sh in "128 128 80 80" "192 128 80 80" "384 128 80 80" "256 128 40 40"; do
    set -- $cfg $sh
    echo -n "KS=$1 NS=$2  "
    YL_MICRO_B=64 YL_CONV_GEMM_BLOCKS=1 python tools/conv_micro.py $3 $4 1 1 $5 $6 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
make -C yolo-lite_amd/csrc -B conv_gemm.o > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1
