#!/bin/bash
# Development aid: big-filter layers with chunked slabs (default) vs the largest MT whose whole filter fits the slab.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "256 64 3 1 20 20" "128 64 3 1 40 40" "128 256 3 2 40 40" "128 128 3 2 80 80" "128 128 3 2 40 40" "64 64 3 2 160 160"; do
  YL_CONV_CHUNK=1 python3 $R/tools/conv_micro.py $cfg 2>&1 | grep dbg | sed 's/^/chunk /'
  YL_CONV_CHUNK=0 python3 $R/tools/conv_micro.py $cfg 2>&1 | grep dbg | sed 's/^/fit   /'
done
