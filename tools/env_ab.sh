#!/bin/bash
# Development aid: images/s of the default bench under different environment settings, alternating on one box.
#   bash tools/env_ab.sh "A=1" "A=2 B=3" ...     (an empty string = the defaults); REPS (default 2) rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/env_ab
mkdir -p $OUT
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v timeout -k 10 200 python3 $R/bench.py --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/v${i}_$rep.json 2> $OUT/v${i}_$rep.err
    echo "rep $rep [$v]: $(python3 -c "import json; d=json.loads(open('$OUT/v${i}_$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -n 1)"
  done
done
