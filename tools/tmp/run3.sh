cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -x -q -k sppf 2>&1 | tail -5 || exit 1
bash tools/ab_env.sh YL_FUSE_SPPF 0 1 2 --steps 500 --warmup 20
python bench.py --no-cpu-baseline --no-other-configs --steps 200 --profile-out gpurun_out/sppf_per_op.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/sppf_per_op.json'))
rows=d['rows'] if isinstance(d,dict) and 'rows' in d else d
for r in rows:
    if r['name'].startswith(('model.8','model.9','model.10')): print(r['name'], r['kernel'], round(r['ms']*1e3,1))
print('sum', round(sum(r['ms'] for r in rows)*1e3,1), len(rows))
PY
rm -f yolo-lite_amd/csrc/sppf_resident.o yolo-lite_amd/csrc/engine.o
make -C yolo-lite_amd/csrc EXTRA=-DYL_STAMPS > /dev/null 2>&1
python tools/res_stamps.py sppf 32 20 20 5 1
