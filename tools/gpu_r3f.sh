cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
echo "== fused tests"; timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -12
echo "== whole suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12
echo "== probe"; timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0"
echo "== driver line"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $O/r3f_driver_line.json | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s (events %s) frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('avg_launch_ms_hip_events'), r.get('frac')))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_e; timeout 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/p_e -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p_e.log 2>&1
cp $(find /tmp/p_e -name "*kernel_trace.csv" | head -1) $O/r3f_kernel_trace.csv
cp $(find /tmp/p_e -name "*hip_api_trace.csv" | head -1) $O/r3f_hip_trace.csv
cd $GRAFT_REPO_ROOT
for k in 64 1024; do echo "== K=$k"; timeout 300 python bench.py --steps $k --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac')))"; done
for c in cfg3 cfg5; do echo "== $c"; timeout 300 python bench.py --config $c --steps 256 --warmup 32 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac')))"; done
