O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s bound %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac'), r.get('bound')))
"; }
echo "== new tests"; timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "native or aborted" 2>&1 | tail -15
for c in cfg3 cfg5; do
echo "== $c native"; timeout 300 python bench.py --config $c --no-cpu-baseline 2>$O/s29_$c.err | summ
echo "== $c chunked"; RIAB_NO_NATIVE=1 timeout 300 python bench.py --config $c --no-cpu-baseline 2>>$O/s29_$c.err | summ
echo "== $c native K=20"; timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/s29_$c.err | summ
echo "== $c chunked K=20"; RIAB_NO_NATIVE=1 timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/s29_$c.err | summ
done
tail -5 $O/s29_cfg3.err
