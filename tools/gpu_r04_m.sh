O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
RIAB_HIP_LIB=tools/exp/libpipe_prof.so timeout 300 python tools/pipe_profile.py 20 2>&1 | grep -v amdgpu.ids | head -8
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys
sys.argv = ["r04_probe.py", "20", "300"]
sys.path.insert(0, "tools")
import r04_probe as P
for _ in range(3):
    P.measure("C gate=reserved (default)", {"RIAB_GATE": "reserved"})
PY
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver cmd: %.4f G  region %.2f us (min %.2f)  frac %.4f  kernel frac %.4f / %.4f' % (o['value']/1e9, 1e3*o['timed_region_ms']['median'], 1e3*o['timed_region_ms']['min'], o['frac_whole_path'], o['roofline']['frac'], o['roofline'].get('frac_device_clock',0)))"; done
