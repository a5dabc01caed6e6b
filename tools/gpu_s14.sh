O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s bound %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac'), r.get('bound')))
"; }
for i in 1 2; do echo "== driver cmd run $i"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | summ; done
echo "== K=1024"; timeout 300 python bench.py --steps 1024 --warmup 5 --no-cpu-baseline 2>/dev/null | summ
echo "== cfg3"; timeout 300 python bench.py --config cfg3 --no-cpu-baseline 2>$O/s14_cfg3.err | tee $O/s14_cfg3.json | summ
echo "== cfg5"; timeout 300 python bench.py --config cfg5 --no-cpu-baseline 2>$O/s14_cfg5.err | tee $O/s14_cfg5.json | summ
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
