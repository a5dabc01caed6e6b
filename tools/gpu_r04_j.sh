O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== two processes on one GPU, 20000 regions of simulate(20) each, reserving shape (no started gate)"
RIAB_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --repeats 20000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
o=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('value %.3f G  region median %.1f us  per rank %s  diagnostics %s  pipeline %s' % (o['value']/1e9, 1e3*o['timed_region_ms']['median'], o['timed_region_ms_per_rank'], o['diagnostics'], o['pipeline']))" | tee $O/r04j_shared_gpu_soak.txt
echo "== stress 40000"; timeout 1800 python tools/fused_stress.py 40000 2>&1 | grep -v amdgpu.ids | tee $O/r04j_fused_stress_40000.txt
