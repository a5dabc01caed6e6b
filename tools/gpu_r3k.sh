cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
echo "== whole suite"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
echo "== exp_bench"; timeout 120 ./tools/exp_bench 2>&1 | tail -12 | tee $O/r3k_exp_bench.txt
echo "== bench line"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | tee $O/r3k_driver_line.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.4g median %.4f kernel %s events %s frac %s' % (d['value'], d['timed_region_ms']['median'], r['avg_launch_ms'], r.get('avg_launch_ms_hip_events'), r['frac']))
for k,v in d.get('secondary',{}).items(): print(k, {x:v.get(x) for x in ('value','ms_per_step','error')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('bound'))"
echo "== cfg3 / cfg5 with the general ray stage (A/B)"
for c in cfg3 cfg5; do RIAB_NO_BVC_BOX=1 timeout 300 python bench.py --config $c --steps 256 --warmup 32 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$c nobox value %.4g' % d['value'])"; done
for c in cfg3 cfg5; do RIAB_NO_BVC_WINDOWS=1 timeout 300 python bench.py --config $c --steps 256 --warmup 32 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$c nowindows value %.4g' % d['value'])"; done
echo "== host split"; timeout 300 python tools/host_split.py 20 2>&1 | grep -v Warn | tail -4
