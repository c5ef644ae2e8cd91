#!/bin/bash
# Development aid: bench.py throughput against batches in flight and hardware queues.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for s in ${STREAMS:-2 3 4 6 8}; do
  for q in ${QUEUES:-8 16}; do
    GPU_MAX_HW_QUEUES=$q python3 $R/bench.py --steps 60 --warmup 12 --streams $s --no-cpu-baseline 2>/dev/null > /tmp/sw.json
    python3 -c "
import json
d=json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1]); print('streams $s queues $q', d['value'], d['ms_per_step'])"
  done
done
