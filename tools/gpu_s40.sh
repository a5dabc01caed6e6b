cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac')))
"; }
timeout 900 python -m pytest tests -m gpu -x -q -k "bvc or BVC or worlds or boundary or cfg3 or windows or field_of_view" 2>&1 | tail -4
for i in 1 2; do
echo "== cfg3"; timeout 300 python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | summ
echo "== cfg3 no windows"; RIAB_NO_BVC_WINDOWS=1 timeout 300 python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | summ
echo "== cfg5"; timeout 300 python bench.py --config cfg5 --no-cpu-baseline 2>/dev/null | summ
done
