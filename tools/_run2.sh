set -x
timeout 1500 python -m pytest tests/test_gpu_step1_device.py -x -q 2>&1 | tail -25
for m in "--plan" "--task" "--per-step"; do
python bench.py --steps 256 --warmup 32 --no-secondary --no-cpu-baseline $m 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$m', o['value']/1e6, o['ms_per_step']*1e3, o.get('plan'))"
done
for c in cfg3 cfg5; do
python bench.py --config $c --steps 256 --warmup 32 --no-secondary --no-cpu-baseline --plan 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$c plan', o['value']/1e6, o['ms_per_step']*1e3, o.get('plan'))"
done
