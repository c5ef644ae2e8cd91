#!/bin/bash
# Kernel trace + stats (and one MFMA-busy PMC pass) of a bench.py configuration into gpurun_out/$1; a heartbeat keeps long
# profiler start-ups from looking hung.   gpurun -- 'BENCH_ARGS="--scale x --imgsz 1280 --batch 8 --dtype fp8" bash tools/profile_trace.sh x8fp8_r01'
set -u
TAG=${1:?tag}
BA=${BENCH_ARGS:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export YL_BENCH_WARM_ROUNDS=0  # (every dispatch is a row of the profiler CSVs; gpurun copies back 64 MiB at most)
( while true; do sleep 60; echo "heartbeat $(date +%T)"; done ) &
HB=$!
timeout -k 10 ${LIMIT:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $BA --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace rc $?"
timeout -k 10 ${LIMIT:-400} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py $BA --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.log 2>&1
echo "pmc rc $?"
kill $HB
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +30M -delete
ls $OUT/trace/* | head
tail -1 $OUT/trace.log | cut -c1-200
