cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
echo "== whole suite"; timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6
echo "== bench line"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | tee $O/r3n_driver_line.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.4g median %.4f kernel %s events %s frac %s traffic %s' % (d['value'], d['timed_region_ms']['median'], r['avg_launch_ms'], r.get('avg_launch_ms_hip_events'), r['frac'], r.get('traffic')))
for k,v in d.get('secondary',{}).items(): print(k, {x:v.get(x) for x in ('value','ms_per_step','error')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('bound'))"
