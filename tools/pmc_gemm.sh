#!/bin/bash
# Development aid: SQ counters of conv_gemm_kernel on one conv_micro configuration (CFG="cin cout 1 1 H W", B, W8).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export YL_CONV_GEMM=1 YL_CONV_GEMM_BLOCKS=1 YL_CONV_GEMM_W8=${W8:-0} YL_MICRO_B=${B:-8}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pg$i
  rocprofv3 --pmc $set -d /tmp/pg$i --output-format csv -- python3 $R/tools/conv_micro.py ${CFG:-768 768 1 1 160 160} > /tmp/pg$i.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pg$i conv_gemm
done
