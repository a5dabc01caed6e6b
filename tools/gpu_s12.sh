O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== auto loop + cfg1"; timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -k "unchanged_reference_loop or cfg1" 2>&1 | tail -30 | tee $O/s12_auto.log
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max']))
"; }
echo "== per-step auto"; timeout 300 python bench.py --no-cpu-baseline --per-step --steps 600 --warmup 50 2>/dev/null | summ
echo "== per-step eager"; RIAB_NO_AUTO_PLAN=1 timeout 300 python bench.py --no-cpu-baseline --per-step --steps 600 --warmup 50 2>/dev/null | summ
echo "== plan"; timeout 300 python bench.py --no-cpu-baseline --plan --steps 600 --warmup 50 2>/dev/null | summ
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/s12_gpu_tests.log
