cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac')))
"; }
run() { echo "== $1 K=$2"; env $1 timeout 300 python bench.py --gpus 1 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | summ; }
for rep in 1 2 3; do
run "X=0" 20
run "RIAB_EXP_NO_GATE=1" 20
run "RIAB_EXP_NO_GATE=1 RIAB_EXP_PREFIX=4" 20
run "RIAB_EXP_PREFIX=4" 20
done
run "X=0" 64
run "RIAB_EXP_NO_GATE=1 RIAB_EXP_PREFIX=4" 64
echo "== fused tests with both"; RIAB_EXP_NO_GATE=1 RIAB_EXP_PREFIX=4 timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -2
