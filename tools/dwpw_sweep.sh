mkdir -p gpurun_out/dw
for cfg in "64 80 80 80" "80 80 80 80" "128 40 40 80"; do
 export YL_GROUP_LEVELS=0  # (the grouped class tails have no LDS-staged form)
for v in "YL_DWPW_TILE=0 YL_DWPW_NT=2" "YL_DWPW_TILE=0 YL_DWPW_NT=1" "YL_DWPW_TILE=1 YL_DWPW_NT=2" "YL_DWPW_TILE=1 YL_DWPW_NT=1" "YL_DWPW_TILE=1 YL_DWPW_NT=1 YL_DWPW_TILE_WPC=2" "YL_DWPW_TILE=1 YL_DWPW_NT=1 YL_DWPW_TILE_WPC=4" "YL_DWPW_TILE=1 YL_DWPW_NT=2 YL_DWPW_TILE_WPC=1"; do
  echo -n "$v :: "; env $v timeout -k 10 60 python3 tools/dw_micro.py $cfg 2>&1 | grep dwconv
 done
done
