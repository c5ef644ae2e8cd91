#!/bin/bash
# Development aid: images/s against (GPU_MAX_HW_QUEUES, batches in flight), single-GPU path or (MODE=sharded) the
# RCCL scatter / gather path with a process group of one rank.  Usage: [MODE=sharded] bash tools/queue_sweep.sh "8 8" "16 16" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/queue_sweep
mkdir -p $OUT
for q in "$@"; do
  set -- $q
  if [ "$MODE" = "sharded" ]; then export YL_BENCH_FORCE_SHARDED=1; fi
  GPU_MAX_HW_QUEUES=$1 timeout -k 10 200 python3 $R/bench.py --streams $2 --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-other-configs \
    > $OUT/${MODE:-single}_$1_$2.json 2> $OUT/${MODE:-single}_$1_$2.err
  echo "rc=$? ${MODE:-single} queues $1 streams $2: $(python3 -c "import json,sys; d=json.loads(open('$OUT/${MODE:-single}_$1_$2.json').read()); print(d['value'], d['ms_per_step'], d.get('value_resident'))" 2>&1 | tail -n 1)"
done
