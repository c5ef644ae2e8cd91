#!/bin/bash
# Development aid: bench.py throughput for several lane maps (YL_LANES = box0 cls0 box1 cls1 box2 cls2; side lanes are opt-in
# since round 2: export YL_SIDE_LANES=1 -- and read DESIGN.md section 4 before trusting a forced-graph-queue result).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for m in ${MAPS:-010000 000000 010100 010101 110000 010001 000100}; do
  YL_LANES=$m python3 $R/bench.py --steps 60 --warmup 12 --no-cpu-baseline 2>/dev/null > /tmp/ls.json
  python3 -c "
import json
d=json.loads(open('/tmp/ls.json').read().strip().splitlines()[-1]); print('lanes $m', d['value'], d['ms_per_step'], d['latency_ms_single_stream'])"
done
