cd "$(dirname "$0")/.."
T=/tmp/yl_tuning
rm -rf $T && mkdir -p $T/yl && cp -r yolo-lite_amd/csrc $T/yl/csrc && cp -r include $T/include
( cd $T/yl/csrc && rm -f *.o && make -j16 EXTRA=-DYL_TUNING > $T/build.log 2>&1 ) || { tail -5 $T/build.log; exit 1; }
for dbg in 0 4 6 7 0; do
  YL_AMD_LIB=$T/yl/libyl_amd.so YL_CONV_DBG=$dbg timeout -k 10 200 python bench.py --steps 50 --no-cpu-baseline --no-other-configs --profile-out gpurun_out/abl_$dbg.json > gpurun_out/abl_b.json 2> gpurun_out/abl.err || { tail -3 gpurun_out/abl.err; continue; }
  python - $dbg <<'PY'
import json,sys
rows=json.load(open(f"gpurun_out/abl_{sys.argv[1]}.json"))
t={r['name'][:20]:round(r['ms']*1e3,2) for r in rows if 'cv2.0.0' in r['name'] or 'cs' in r['kernel'] or r['name'] in ('model.3+cv1','model.7')}
print("YL_CONV_DBG="+sys.argv[1], t, flush=True)
PY
done
