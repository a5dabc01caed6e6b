for c in cfg3 cfg5; do
for e in 0 1; do
RIAB_NO_SIDE_POPS=$e python bench.py --config $c --steps 1024 --warmup 32 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$c no_side_pops=$e', round(o['value']/1e6,1), 'M', round(o['ms_per_step']*1e3,2), 'us/step', o['roofline'].get('kernel'), o['roofline'].get('avg_launch_ms'), o['roofline'].get('frac'))"
done; done
timeout 900 python -m pytest tests/test_gpu_bench_length.py tests/test_gpu_fused.py -x -q 2>&1 | tail -4
