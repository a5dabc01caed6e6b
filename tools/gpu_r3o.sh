cd $GRAFT_REPO_ROOT
echo "== fused tests"; timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -4
echo "== host split"; timeout 300 python tools/host_split.py 20 2>&1 | grep -v Warn | tail -4
echo "== probe"; timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0"
echo "== bench"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.4g median %.4f min %.4f kernel %s events %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r['avg_launch_ms'], r.get('avg_launch_ms_hip_events'), r['frac']))"
