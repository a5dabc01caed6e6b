cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac')))
"; }
echo "== driver"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | summ
echo "== default"; timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | summ
echo "== per-step"; timeout 300 python bench.py --per-step --steps 1024 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== per-step eager"; RIAB_NO_AUTO_PLAN=1 timeout 300 python bench.py --per-step --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== plan"; timeout 300 python bench.py --plan --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== task"; timeout 300 python bench.py --task --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== cfg4"; timeout 300 python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | summ
