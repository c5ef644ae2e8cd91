#!/bin/bash
# Collects the round's profile on the GPU box into gpurun_out/$1 (kernel trace + stats, HBM counters in
# separate passes as MI355X_MICROARCH.md prescribes, the un-profiled bench line and the per-op table).
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01'
# then:  python tools/summarize_profile.py gpurun_out/r01 profiles r01
set -u
TAG=${1:-r01}
BA=${BENCH_ARGS:-}   # e.g. BENCH_ARGS="--scale x --imgsz 1280 --batch 8 --dtype fp8" with a tag that does not start with r
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "bench $(date +%T)"
timeout -k 10 ${LIMIT:-500} python3 $R/bench.py $BA --profile-out $OUT/per_op.json > $OUT/bench.json 2> $OUT/bench.err
# the profiled passes run the THROUGHPUT profile of the program only (the latency profile's extra launches -- the predictor's per-call
# engine, the bench's one_batch_profile engine -- would mix another population into the per-kernel averages)
export YL_BENCH_ONE_BATCH_PROFILE=0 YL_PREDICT_ONE_BATCH_PROFILE=0
# (no extra warm-up rounds under the profiler: every dispatch is a row of the counter CSVs, and gpurun copies back 64 MiB at most)
export YL_BENCH_WARM_ROUNDS=0
# (one batch in flight: a kernel's duration is its own -- with 20 batches in flight, all started within 0.2 ms by the issue threads,
# the same kernels' wall durations stretch 3 x (conv1x1: 13 -> 44 us) while the throughput is unchanged)
echo "pass --stats $(date +%T)"
timeout -k 10 ${LIMIT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $BA --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-other-configs > $OUT/trace.log 2>&1
python3 $R/tools/reduce_pmc.py $OUT/trace
echo "pass FETCH_SIZE $(date +%T)"
timeout -k 10 ${LIMIT:-300} rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $BA --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/pmc_fetch.log 2>&1
python3 $R/tools/reduce_pmc.py $OUT/pmc_fetch
echo "pass WRITE_SIZE $(date +%T)"
timeout -k 10 ${LIMIT:-300} rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $BA --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/pmc_write.log 2>&1
python3 $R/tools/reduce_pmc.py $OUT/pmc_write
# independent read-side count (round 3): L2 misses x 128-byte lines, and the fabric read requests with their 32-byte share
# (SKIP_MISS=1: without this cross-check pass -- its CSV is three counters wide, and gpurun copies back 64 MiB at most: the
# batch-64 / 1280x1280 configurations do not fit with it)
[ "${SKIP_MISS:-0}" = "1" ] || echo "pass TCC_MISS_sum $(date +%T)"
[ "${SKIP_MISS:-0}" = "1" ] || timeout -k 10 ${LIMIT:-300} rocprofv3 --pmc TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc_miss -- python3 $R/bench.py $BA --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/pmc_miss.log 2>&1
[ "${SKIP_MISS:-0}" = "1" ] || python3 $R/tools/reduce_pmc.py $OUT/pmc_miss
# matrix-pipe occupancy per kernel: busy cycles summed over the 1024 SIMDs / (1024 x kernel cycles); one batch at a time
# (--streams 1) so that GRBM_GUI_ACTIVE of a dispatch is that kernel's own duration.  (Round 4: this pass crashed the profiled process
# in 4 of 6 attempts until the hot path's captured hipMemsetAsync was replaced by a kernel -- csrc/nms.hip zero_i32 -- and ran clean since.)
echo "pass SQ_VALU_MFMA_BUSY_CYCLES $(date +%T)"
timeout -k 10 ${LIMIT:-300} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py $BA --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-other-configs > $OUT/pmc_mfma.log 2>&1
python3 $R/tools/reduce_pmc.py $OUT/pmc_mfma
# issue / wait / LDS / texture-addresser counters for the top kernels (round 6; SQ block: 8 slots per pass), one batch in flight
if [ "${SKIP_SQ:-0}" != "1" ]; then
echo "pass SQ issue counters $(date +%T)"
timeout -k 10 ${LIMIT:-300} rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 $R/bench.py $BA --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-other-configs > $OUT/pmc_sq1.log 2>&1
python3 $R/tools/reduce_pmc.py $OUT/pmc_sq1
echo "pass SQ LDS / TA counters $(date +%T)"
timeout -k 10 ${LIMIT:-300} rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM TA_BUSY_avr --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $BA --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-other-configs > $OUT/pmc_sq2.log 2>&1
python3 $R/tools/reduce_pmc.py $OUT/pmc_sq2
fi
# (every pass under its own time limit and announced on stdout: a pass that hangs -- one did, after a profiler-side memory fault two
# seconds into the process -- must not take the call with it)
# keep only the small CSVs
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT $OUT/trace/* | head -30
tail -1 $OUT/bench.json
