for cfg in "256 256 1 1 20 20" "128 128 1 1 40 40" "256 128 1 1 40 40"; do
  for d in 0 1 2 3 4 7 15; do
    echo -n "cfg [$cfg] dbg $d: "; YL_CONV_DBG=$d timeout -k 10 60 python3 tools/conv_micro.py $cfg 2>&1 | grep -v amdgpu | tail -n 1
  done
done
