cd /tmp && export TMPDIR=/tmp
for c in cfg5; do
rm -rf /tmp/p_$c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 1024 --warmup 32 --no-cpu-baseline --no-secondary > /tmp/p_$c.log 2>&1
f=$(find /tmp/p_$c -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r06_${c}_kernel_stats.csv
echo == $c; cut -d, -f1-7 $f | cut -c1-230 | head -8; tail -1 /tmp/p_$c.log | cut -c1-300
done
