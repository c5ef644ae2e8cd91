#!/bin/bash
# Development aid: SQ counters of the filter-in-registers conv kernel on one conv_micro configuration.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export YL_CONV_REG=1 YL_CONV_REG_MIN=1 YL_CONV_REG_NT=${NT:-4}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pr$i
  rocprofv3 --pmc $set -d /tmp/pr$i --output-format csv -- python3 $R/tools/conv_micro.py ${CFG:-64 64 3 1 80 80} > /tmp/pr$i.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pr$i conv
done
