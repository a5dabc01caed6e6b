# rocprofv3 kernel stats of bench.py --config $1; summary lands in gpurun_out/$1_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python $GRAFT_REPO_ROOT/bench.py --config $1 --no-cpu-baseline > /tmp/prof_$1.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp $(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/$1_kernel_stats.csv
grep '^{"metric"' /tmp/prof_$1.log | tail -1 | cut -c80-150
