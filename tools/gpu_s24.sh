O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac')))
"; }
echo "== fused tests"; timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -2
for k in 20 128 256 257 512 1024 1024 4096; do
echo "== K=$k"; timeout 300 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null | summ
done
for pm in 64 128; do echo "== POLL_MAX=$pm K=256"; RIAB_STREAM_POLL_MAX=$pm timeout 300 python bench.py --gpus 1 --steps 256 --warmup 5 --no-cpu-baseline 2>/dev/null | summ; done
echo "== POLL_MAX=64 K=128"; RIAB_STREAM_POLL_MAX=64 timeout 300 python bench.py --gpus 1 --steps 128 --warmup 5 --no-cpu-baseline 2>/dev/null | summ
