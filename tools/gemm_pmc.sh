#!/bin/bash
# Memory-side counters of one conv_gemm_kernel shape (separate --pmc passes): UTCL1 (TLB) misses, TCP -> TCC read latency, L2 hit rate.
# Usage: bash tools/gemm_pmc.sh B cin cout H W  (writes gpurun_out/gemm_pmc/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gemm_pmc
rm -rf $OUT; mkdir -p $OUT
export YL_MICRO_B=$1 YL_CONV_GEMM_BLOCKS=1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/conv_micro.py $2 $3 1 1 $4 $5 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 6):
    fs = glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "conv_gemm" in k:
            print(i, k, {c: round(v / cnt[(k, c)], 1) for c, v in d.items()})
PY
find $OUT -name "*.db" -delete
