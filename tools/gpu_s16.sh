O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac')))
"; }
for f in 0 2 0 2; do
for k in 20 1024; do
echo "== fillers $f K=$k"; RIAB_EXP_FILLERS=$f timeout 300 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null | summ
done; done
for f in 0 2; do echo "== fillers $f K=4096"; RIAB_EXP_FILLERS=$f timeout 300 python bench.py --gpus 1 --steps 4096 --warmup 5 --no-cpu-baseline 2>/dev/null | summ; done
echo "== fused tests fillers 2"; RIAB_EXP_FILLERS=2 timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
