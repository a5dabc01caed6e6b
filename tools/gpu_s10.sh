O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== ops tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -30 | tee $O/s10_ops_tests.log
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac')))
"; }
echo "== per-step"; timeout 300 python bench.py --no-cpu-baseline --per-step --steps 300 --warmup 20 2>/dev/null | summ
echo "== plan"; timeout 300 python bench.py --no-cpu-baseline --plan --steps 300 --warmup 20 2>/dev/null | summ
for k in 20 1024; do echo "== K=$k"; timeout 300 python bench.py --no-cpu-baseline --steps $k --warmup 5 2>/dev/null | summ; done
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee $O/s10_gpu_tests.log
