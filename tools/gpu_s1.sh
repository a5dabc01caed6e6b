# GPU session 1 (round 2): fused-pipeline tests, the driver's bench command, A/B sweeps.  Every step bounded.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== fused tests"; timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -25 | tee $O/s1_fused_tests.log
echo "== driver cmd"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/s1_bench_k20.json 2> $O/s1_bench_k20.err; tail -c 3000 $O/s1_bench_k20.json
echo "== default"; timeout 300 python bench.py --no-cpu-baseline > $O/s1_bench_default.json 2>> $O/s1_bench_k20.err; tail -c 2500 $O/s1_bench_default.json
for v in "RIAB_NO_FUSED=1" "RIAB_STREAM_WGS_PER_CU=5" "RIAB_STREAM_WGS_PER_CU=6" "RIAB_STREAM_WGS_PER_CU=8" "RIAB_STREAM_GPI=1" "RIAB_STREAM_GPI=4" "RIAB_STREAM_MODE=1"; do
  for k in 20 1024; do
    echo "== $v K=$k"; env $v timeout 300 python bench.py --no-cpu-baseline --steps $k --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s  timeouts %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac'), d['diagnostics'].get('pipeline_timeouts')))
"
  done
done 2>&1 | tee $O/s1_sweep.log
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/s1_gpu_tests.log
