# round 3, call A: sanity of ABI v4 (gate only when the stream is busy), the K=20 host/GPU split, a trace
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
echo "== fused + ops tests"; timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3
echo "== probe"; timeout 600 python tools/k20_probe.py 20 300 2>&1 | grep -v Warning | tail -12
echo "== driver line"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $O/r3a_driver_line.json | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac')))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_a; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_a -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p_a.log 2>&1
cp $(find /tmp/p_a -name "*kernel_trace.csv" | head -1) $O/r3a_driver_kernel_trace.csv
grep '^{"metric"' /tmp/p_a.log | tail -1 > $O/r3a_driver_line_under_rocprof.json
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3a_driver_kernel_trace.csv')))
rows=[r for r in rows if 'agent_step' in r['Kernel_Name'] or 'rate_kernel' in r['Kernel_Name'] or 'gate' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=None
for r in rows[-9:]:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    if 'agent_step' in r['Kernel_Name']: t0=s
    print('%-40s start %+8.2f us  dur %7.2f us' % (r['Kernel_Name'][:40], (s-(t0 or s))/1e3, (e-s)/1e3))
PY
