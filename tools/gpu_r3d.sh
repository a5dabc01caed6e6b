cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
echo "== fused tests"; timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -5
echo "== whole suite"; timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12
echo "== standalone trajectory kernel"
timeout 120 python tools/traj_probe.py 2>&1 | grep "T="
RIAB_TRAJ2=1 timeout 120 python tools/traj_probe.py 2>&1 | grep "T="
for b in 3 31; do RIAB_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/libt4_$b.so timeout 120 python tools/traj_probe.py 2>&1 | grep "T="; done
echo "== probe"; timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_b; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p_b.log 2>&1
cp $(find /tmp/p_b -name "*kernel_trace.csv" | head -1) $O/r3d_driver_kernel_trace.csv
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3d_driver_kernel_trace.csv')))
rows=[r for r in rows if 'agent_step' in r['Kernel_Name'] or 'traj4' in r['Kernel_Name'] or 'rate_kernel' in r['Kernel_Name'] or 'gate' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=None
for r in rows[-9:]:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    if 'traj4' in r['Kernel_Name'] or 'agent_step' in r['Kernel_Name']: t0=s
    print('%-40s start %+8.2f us  dur %7.2f us' % (r['Kernel_Name'][:40], (s-(t0 or s))/1e3, (e-s)/1e3))
PY
