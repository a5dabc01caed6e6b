O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== 8 ranks on one GPU (the launch form and control flow of an 8-GPU line; gloo control plane)"
RIAB_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --secondary-timeout 300 2> $O/r04l_8ranks_err.txt | tail -1 > $O/r04l_8ranks_line.json
python - <<'PY'
import json
o=json.load(open("gpurun_out/r04l_8ranks_line.json"))
print("n_gpus", o["n_gpus"], "value %.3f G" % (o["value"]/1e9), "region median %.1f us" % (1e3*o["timed_region_ms"]["median"]))
print("per rank:", [(x["rank"], round(1e3*x["median"],1)) for x in o["timed_region_ms_per_rank"]])
print("binding:", o["config"]["host_binding_per_rank"][:2], "...")
print("diagnostics", o["diagnostics"], "secondary_error" in o)
for k,v in o.get("secondary",{}).items(): print(k, v.get("value"), v.get("steps"), v.get("error"), v.get("diagnostics",{}).get("pipeline_timeouts"))
PY
tail -3 $O/r04l_8ranks_err.txt
