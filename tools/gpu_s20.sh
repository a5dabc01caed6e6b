O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac')))
"; }
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for k in 20 20 64 256 1024 4096; do
echo "== K=$k"; timeout 300 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null | summ
done
echo "== per-step"; timeout 300 python bench.py --per-step --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== plan"; timeout 300 python bench.py --plan --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
echo "== task"; timeout 300 python bench.py --task --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | summ
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_drv -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p_drv.log 2>&1
cp $(find /tmp/p_drv -name "*kernel_stats.csv" | head -1) $O/s20_driver_kernel_stats.csv
cut -d, -f1-4 $O/s20_driver_kernel_stats.csv | cut -c1-50,90-160 | head -4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_def -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/p_def.log 2>&1
cp $(find /tmp/p_def -name "*kernel_stats.csv" | head -1) $O/s20_default_kernel_stats.csv
cut -d, -f1-4 $O/s20_default_kernel_stats.csv | cut -c1-50,90-160 | head -4
