# round 3, call B: the four-wave trajectory kernel: bit-identity with the single-wave kernel, the suite, K=20 timing
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
echo "== fused tests"; timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -15
echo "== whole suite"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
echo "== probe (4 waves)"; timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0 timing=0 gate=auto\|spin=0 timing=1 gate=auto"
echo "== probe (2 waves)"; RIAB_TRAJ2=1 timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0 timing=0 gate=auto"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_b; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p_b.log 2>&1
cp $(find /tmp/p_b -name "*kernel_trace.csv" | head -1) $O/r3b_driver_kernel_trace.csv
grep '^{"metric"' /tmp/p_b.log | tail -1 > $O/r3b_driver_line_under_rocprof.json
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3b_driver_kernel_trace.csv')))
rows=[r for r in rows if 'agent_step' in r['Kernel_Name'] or 'traj4' in r['Kernel_Name'] or 'rate_kernel' in r['Kernel_Name'] or 'gate' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=None
for r in rows[-8:]:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    if 'traj4' in r['Kernel_Name'] or 'agent_step' in r['Kernel_Name']: t0=s
    print('%-40s start %+8.2f us  dur %7.2f us' % (r['Kernel_Name'][:40], (s-(t0 or s))/1e3, (e-s)/1e3))
PY
cd $GRAFT_REPO_ROOT
for k in 64 1024; do echo "== K=$k"; timeout 300 python bench.py --steps $k --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r.get('avg_launch_ms'), r.get('frac')))"; done
