#!/bin/bash
# Cache-policy bits of conv_gemm_kernel's activation pieces on single-channel-block layers (build switch YL_GEMM_LD_AUX): rebuilds ON THE GPU BOX.
cd "$(dirname "$0")/.."
for aux in ${AUXS:-0 2 0 2}; do
  make -C yolo-lite_amd/csrc -B conv_gemm.o EXTRA="-DYL_GEMM_LD_AUX=$aux" > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  for sh in "64 128 128 80 80" "32 256 64 80 80" "64 192 128 80 80"; do
    set -- $sh
    echo -n "ld aux=$aux B=$1  "
    YL_MICRO_B=$1 YL_CONV_GEMM_BLOCKS=1 python tools/conv_micro.py $2 $3 1 1 $4 $5 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
make -C yolo-lite_amd/csrc -B conv_gemm.o > /dev/null 2>&1 && make -C yolo-lite_amd/csrc > /dev/null 2>&1
