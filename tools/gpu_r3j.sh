cd $GRAFT_REPO_ROOT
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q 2>&1 | tail -8
echo "== probe sc1";  timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0 timing=True"
echo "== probe plain"; RIAB_GATED_PLAIN=1 timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0 timing=True"
echo "== fused tests with plain loads"; RIAB_GATED_PLAIN=1 timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
echo "== bench line"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.4g median %.4f kernel %s events %s frac %s' % (d['value'], d['timed_region_ms']['median'], r['avg_launch_ms'], r.get('avg_launch_ms_hip_events'), r['frac']))
for k,v in d.get('secondary',{}).items(): print(k, {x:v.get(x) for x in ('value','ms_per_step','wall_s','error')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('bound'))"
