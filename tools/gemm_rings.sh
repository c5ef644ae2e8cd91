#!/bin/bash
# Ring shapes of conv_gemm_kernel's 192-channel blocks (build switches YL_GM6_KS / _NS: four waves, YL_GM6W_KS / _NS: eight waves)
# (eight waves: 12 filter pieces per k-step do not divide over 8 waves, so KS = 2 only, and three slices of it exceed the LDS)
# on 384 -> 384 @320x320 batch 8 (YOLO11x 1280's model.2.cv2 / model.4.cv2 / model.16.cv1): rebuilds conv_gemm.o ON THE GPU BOX.
cd "$(dirname "$0")/.."
for cfg in "2 2 2 2" "1 4 2 2" "2 3 2 2" "1 6 2 2" "1 3 2 2" "2 4 2 2"; do
  set -- $cfg
  make -C yolo-lite_amd/csrc -B conv_gemm.o EXTRA="-DYL_GM6_KS=$1 -DYL_GM6_NS=$2 -DYL_GM6W_KS=$3 -DYL_GM6W_NS=$4" > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  for w8 in 0; do
    [ $w8 = 0 ] && echo -n "four waves  KS=$1 NS=$2  " || echo -n "eight waves KS=$3 NS=$4  "
    YL_MICRO_B=8 YL_CONV_GEMM_W8=$w8 YL_CONV_GEMM_BLOCKS=1 python tools/conv_micro.py 384 384 1 1 320 320 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
make -C yolo-lite_amd/csrc -B conv_gemm.o > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1
