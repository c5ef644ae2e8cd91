#!/bin/bash
# Whole-step resource sums: SQ / GRBM counters of EVERY kernel of one bench.py step (dispatches are serialised under PMC
# collection), summed per resource by tools/pmc_step_summary.py.  Tells which on-chip resource the batches-in-flight
# throughput mode can at best be bound by.  Usage (GPU box): bash tools/pmc_step.sh OUTDIR
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/pmc_step}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/ps$i
  rocprofv3 --pmc $set -d /tmp/ps$i --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-other-configs > /tmp/ps$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $R/tools/pmc_step_summary.py /tmp/ps1 /tmp/ps2 /tmp/ps3 > $OUT/pmc_step.txt
cp $OUT/pmc_step.txt /dev/stdout | tail -n 60
