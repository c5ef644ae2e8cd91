#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "== $1"; shift; env "$@" python3 bench.py --steps 300 --no-cpu-baseline $EXTRA 2>/tmp/err.log | python3 -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:b.get(k) for k in ('value','ms_per_step','latency_ms_single_stream','value_sync_per_iteration','api_images_per_sec')}, b['runtime']['graph_branch_probe'])" || tail -3 /tmp/err.log; }
run "serial graph (default now)" A=1
EXTRA="--streams 6" run "serial graph, 6 streams" A=1
EXTRA="--streams 8" run "serial graph, 8 streams" A=1
EXTRA="--streams 4" run "serial graph, 4 streams" A=1
run "side lanes, runtime default queues" YL_SIDE_LANES=1
run "side lanes + forced graph queues (r01, INVALID)" YL_SIDE_LANES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
