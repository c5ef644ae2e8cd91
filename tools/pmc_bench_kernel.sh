#!/bin/bash
# Development aid: SQ counters of one kernel (name substring $1) inside a short bench.py run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-stem2}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pk$i --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > /tmp/pk$i.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pk$i $K
done
