O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for k in 64 256 1024 4096; do
  w=$((k/8)); timeout 600 python bench.py --gpus 1 --steps $k --warmup $w --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=o['roofline']
print('K=%-5d value %.3f G  region %.3f ms  frac_whole %.3f  form %s  kernel frac %.3f (events) %s (clock)' % (o['steps'], o['value']/1e9, o['timed_region_ms']['median'], o['frac_whole_path'], r.get('rate_stage_form'), r['frac'], r.get('frac_device_clock')))"
done | tee $O/r04n_k_sweep.txt
for mode in --per-step --plan --task; do
  timeout 600 python bench.py --gpus 1 --steps 1024 --warmup 128 --no-cpu-baseline --no-secondary $mode 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode value %.1f M  %.2f us per step' % (o['value']/1e6, 1e3*o['ms_per_step']))"
done | tee -a $O/r04n_k_sweep.txt
