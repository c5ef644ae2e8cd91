cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
for v in 0 1600 0 1600; do YL_F8_KORD=$v python bench.py --no-cpu-baseline --no-other-configs --scale x --batch 8 --imgsz 1280 --dtype fp8 --steps 60 --profile-out gpurun_out/r3q/per_op_x_$v.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('f8kord=$v', d['value'], d['ms_per_step'], d['fp8_vs_bf16']['matched_frac'], d['fp8_vs_bf16']['ok'], {k:(round(x['ms'],3),x['launches']) for k,x in d['kernels'].items() if 'f8' in k})
"; done
python - <<'EOP'
import json
a=json.load(open("gpurun_out/r3q/per_op_x_0.json")); b=json.load(open("gpurun_out/r3q/per_op_x_1600.json"))
for x,y in zip(a,b):
    if "f8" in x["kernel"] and abs(x["ms"]-y["ms"])>0.004: print(x["name"][:40], x["kernel"], round(x["ms"]*1e3,1), "->", round(y["ms"]*1e3,1))
EOP
