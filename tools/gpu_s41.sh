cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac')))
"; }
for i in 1 2 3; do
echo "== cfg3 paired"; timeout 300 python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | summ
echo "== cfg3 single"; RIAB_EXP_NO_RAY_PAIRS=1 timeout 300 python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | summ
done
