timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wall_grid or motion or sixty or cfg1 or rollout" 2>&1 | tail -5
for c in cfg3_64w; do
for e in 1 0; do
RIAB_NO_WALL_GRID=$e python bench.py --config $c --steps 1024 --warmup 32 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$c no_grid=$e', round(o['value']/1e6,1), 'M', round(o['ms_per_step']*1e3,2), 'us/step')"
RIAB_NO_WALL_GRID=$e python bench.py --config $c --steps 256 --warmup 32 --no-secondary --no-cpu-baseline --plan 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$c plan no_grid=$e', round(o['value']/1e6,1), 'M', round(o['ms_per_step']*1e3,2), 'us/step')"
done; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_g; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_g -- python $GRAFT_REPO_ROOT/bench.py --config cfg3_64w --steps 1024 --warmup 32 --no-cpu-baseline --no-secondary > /tmp/p_g.log 2>&1
f=$(find /tmp/p_g -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r06_cfg3_64w_grid_kernel_stats.csv; cut -d, -f1-4 $f | cut -c1-150 | head -5
