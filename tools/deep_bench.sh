#!/bin/bash
# dwpw_deep_kernel against dwpw_kernel on the first class-branch launch of each Detect level (tools/dw_micro.py: C H W cout),
# alternating; then end to end.
cd "$(dirname "$0")/.."
for cfg in "256 20 20 80" "128 40 40 80"; do
  for rep in 1 2; do
    for deep in 0 1; do
      echo -n "cfg=$cfg deep=$deep :: "; YL_DWPW_SMALL_NT1=1024 YL_DWPW_DEEP=$deep timeout -k 10 120 python3 tools/dw_micro.py $cfg 2>&1 | grep -i "dw" | tail -1
    done
  done
done
