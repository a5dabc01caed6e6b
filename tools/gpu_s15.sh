O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s bound %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac'), r.get('bound')))
"; }
for c in cfg3 cfg5; do
echo "== $c streams per population"; timeout 300 python bench.py --config $c --no-cpu-baseline 2>$O/s15_$c.err | summ
echo "== $c one rate stream"; RIAB_ONE_RATE_STREAM=1 timeout 300 python bench.py --config $c --no-cpu-baseline 2>>$O/s15_$c.err | summ
done
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
