O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== fused tests"; timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -15 | tee $O/s3_fused_tests.log
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s  timeouts %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac'), d['diagnostics'].get('pipeline_timeouts')))
"; }
for v in "X=0" "RIAB_STREAM_WGS_PER_CU=6" "RIAB_STREAM_WGS_PER_CU=8" "RIAB_STREAM_GPI=4" "RIAB_STREAM_GPI=16" "RIAB_STREAM_MODE=1"; do
  for k in 20 128 1024; do
    echo "== $v K=$k"; env $v timeout 300 python bench.py --no-cpu-baseline --steps $k --warmup 5 2>/dev/null | summ
  done
done 2>&1 | tee $O/s3_sweep.log
echo "== rocprof trace K=1024 (fused default)"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1024 --warmup 5 --repeats 3 > /tmp/prof1.log 2>&1
grep '^{"metric"' /tmp/prof1.log | summ
cp $(find /tmp/prof1 -name "*kernel_trace.csv" | head -1) $O/s3_k1024_kernel_trace.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /tmp/prof2.log 2>&1
grep '^{"metric"' /tmp/prof2.log | summ
cp $(find /tmp/prof2 -name "*kernel_trace.csv" | head -1) $O/s3_k20_kernel_trace.csv
cd $GRAFT_REPO_ROOT
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/s3_gpu_tests.log
