O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac')))
"; }
echo "== velocity test"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "velocity" 2>&1 | tail -3
echo "== cfg4"; timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | summ
echo "== cfg4 K=20"; timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | summ
echo "== 2 ranks on one GPU, driver form"; timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
