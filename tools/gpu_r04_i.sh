O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2 3; do
for steps in 256 1024; do
for ns in 100000 100; do
  RIAB_FORM_STEP_NS=$ns timeout 300 python bench.py --config cfg3 --steps $steps --warmup 32 --no-cpu-baseline --repeats 7 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg3 steps %d STEP_NS=%-7s value %.1f M  region %.3f ms (min %.3f)' % (o['steps'], '$ns', o['value']/1e6, o['timed_region_ms']['median'], o['timed_region_ms']['min']))"
done; done; done 2>&1 | tee $O/r04i_cfg3_forms.txt
