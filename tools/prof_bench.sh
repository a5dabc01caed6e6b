# rocprofv3 kernel stats + trace of the default bench command; summaries land in gpurun_out/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $BENCH_ARGS > /tmp/prof_bench.log 2>&1
grep '^{"metric"' /tmp/prof_bench.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_line.json
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/cfg2_kernel_stats.csv
cp $(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/cfg2_kernel_trace.csv
head -4 $GRAFT_REPO_ROOT/gpurun_out/cfg2_kernel_stats.csv | cut -c1-200
