O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac')))
"; }
run() { echo "== $1 K=$2"; env $1 timeout 300 python bench.py --gpus 1 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | summ; }
for rep in 1 2; do
for k in 20 128; do
run "X=0" $k
run "RIAB_GATED_CPB_MULT=2" $k
run "RIAB_GATED_CPB_MULT=4" $k
run "RIAB_GATED_PREPOLL=1" $k
run "RIAB_PUB_EAGER=1" $k
run "RIAB_PUB_EAGER=2" $k
run "RIAB_GATED_CPB_MULT=2 RIAB_GATED_PREPOLL=1 RIAB_PUB_EAGER=2" $k
done; done
echo "== fused tests with all three"; RIAB_GATED_CPB_MULT=2 RIAB_GATED_PREPOLL=1 RIAB_PUB_EAGER=2 timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -2
