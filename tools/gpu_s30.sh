O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('value %.4g  median_ms %.4f min %.4f max %.4f  kernel_ms %s frac %s bound %s' % (d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], d['timed_region_ms']['max'], r.get('avg_launch_ms'), r.get('frac'), r.get('bound')))
"; }
for i in 1 2; do
echo "== cfg5 native"; timeout 300 python bench.py --config cfg5 --no-cpu-baseline --repeats 6 2>/dev/null | summ
echo "== cfg5 chunked"; RIAB_NO_NATIVE=1 timeout 300 python bench.py --config cfg5 --no-cpu-baseline --repeats 6 2>/dev/null | summ
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_n -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --repeats 3 > /tmp/p_n.log 2>&1
cp $(find /tmp/p_n -name "*kernel_stats.csv" | head -1) $O/s30_cfg5_native_stats.csv
RIAB_NO_NATIVE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --repeats 3 > /tmp/p_c.log 2>&1
cp $(find /tmp/p_c -name "*kernel_stats.csv" | head -1) $O/s30_cfg5_chunked_stats.csv
