for q in 8 12 16 20; do for s in 6 10 14 18; do
  if [ $s -ge $q ]; then continue; fi
  GPU_MAX_HW_QUEUES=$q YL_BENCH_FORCE_SHARDED=1 timeout -k 10 200 python bench.py --streams $s --no-cpu-baseline --no-other-configs > gpurun_out/sw_${q}_${s}.json 2> gpurun_out/sw_${q}_${s}.err || exit 1
  echo "q=$q s=$s done"
done; done
