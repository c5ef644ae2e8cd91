#!/bin/bash
# Alternating A/B of one environment switch on the default bench (same box): tools/ab_env.sh VAR A B [reps] [bench args...]
cd "$(dirname "$0")/.."
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4
for rep in $(seq $REPS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
post={r:v for r,v in d['kernels'].items() if r.startswith(('decode','nms','dw3x3'))}
print('$VAR=$v', d['value'], 'img/s', 'sync', d.get('value_sync_per_iteration'), 'lat_ms', d.get('latency_ms_single_stream'), 'one_batch_profile_ms', (d.get('one_batch_profile') or {}).get('latency_ms'), 'launches', d.get('launches_per_step'), {k:(round(x['ms']*1e3,1),x['launches']) for k,x in post.items()})
"
  done
done
