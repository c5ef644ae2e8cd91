#!/bin/bash
# PMC passes over one conv_micro configuration (development aid).  usage: pmc_conv.sh "<cin cout k s H W>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ TCP_PENDING_STALL_CYCLES" \
           "TCC_REQ TCC_HIT TCC_MISS TCC_READ" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_TOTAL_CYCLES" \
           "TA_BUFFER_READ_WAVEFRONTS TA_BUFFER_WRITE_WAVEFRONTS TCP_READ_TAGCONFLICT_STALL_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pc$i --output-format csv -- python3 $R/tools/conv_micro.py $1 > /tmp/pc$i.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pc$i conv_kernel
done
