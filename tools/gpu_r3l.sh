cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('$1 value %.4g  median_ms %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], r.get('avg_launch_ms'), r.get('frac')))"; }
echo "== ops test"; timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -x -q -k "operator or place_cells_vs_reference or randomised_worlds" 2>&1 | tail -5
for i in 1 2; do
for c in cfg3 cfg5; do
timeout 300 python bench.py --config $c --steps 256 --warmup 32 --no-cpu-baseline 2>/dev/null | summ "$c box"
RIAB_NO_BVC_BOX=1 timeout 300 python bench.py --config $c --steps 256 --warmup 32 --no-cpu-baseline 2>/dev/null | summ "$c nobox"
done; done
echo "== host split (operator / direct)"; timeout 300 python tools/host_split.py 20 2>&1 | grep -v Warn | tail -4
