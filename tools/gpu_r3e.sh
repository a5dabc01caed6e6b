cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
rm -rf /tmp/p_e; timeout 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/p_e -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p_e.log 2>&1
ls /tmp/p_e/*/ | head
cp $(find /tmp/p_e -name "*kernel_trace.csv" | head -1) $O/r3e_kernel_trace.csv
cp $(find /tmp/p_e -name "*hip_api_trace.csv" | head -1) $O/r3e_hip_trace.csv
python - <<'PY'
import csv,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'
k=list(csv.DictReader(open(O+'r3e_kernel_trace.csv')))
h=list(csv.DictReader(open(O+'r3e_hip_trace.csv')))
print(h[0].keys())
ev=[]
for r in k:
    n=r['Kernel_Name']
    if 'traj4' in n or 'rate_kernel' in n or 'gate' in n:
        ev.append((int(r['Start_Timestamp']),'K+ '+n[:30])); ev.append((int(r['End_Timestamp']),'K- '+n[:30]))
for r in h:
    ev.append((int(r['Start_Timestamp']),'A+ '+r['Function'])); ev.append((int(r['End_Timestamp']),'A- '+r['Function']))
ev.sort()
# last traj4 kernel
t0=[t for t,n in ev if n.startswith('K+ void riab::traj4')][-2]
for t,n in ev:
    if t0-60000 <= t <= t0+140000: print('%+9.2f %s' % ((t-t0)/1e3, n))
PY
