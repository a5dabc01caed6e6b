# What a round ends with, run on the GPU box through gpurun (`gpurun -- 'bash tools/gpu_checks.sh'`): the smoke test,
# the whole -m gpu suite (SKIP_TESTS=1: without it), the example, and one line per bench mode.
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s bound %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac'), r.get('bound')))
        for k,v in d.get('secondary',{}).items(): print('   secondary', k, v.get('value'), (v.get('roofline') or {}).get('frac'), v.get('error'))
"; }
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
[ -n "$SKIP_TESTS" ] || { echo "== all gpu tests"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3; }
echo "== example"; timeout 600 python examples/simple_example.py 2>&1 | tail -6
echo "== driver"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | summ
for k in 64 256 4096; do echo "== K=$k"; timeout 300 python bench.py --steps $k --warmup 16 --no-cpu-baseline --no-secondary 2>/dev/null | summ; done
echo "== default"; timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | summ
echo "== per-step"; timeout 300 python bench.py --per-step --steps 1024 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== plan"; timeout 300 python bench.py --plan --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== task"; timeout 300 python bench.py --task --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== task, one world"; timeout 300 python bench.py --task-world --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
for c in cfg3 cfg4 cfg5; do echo "== $c"; timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | summ; done
