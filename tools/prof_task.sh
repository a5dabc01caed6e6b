cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_task -- python $GRAFT_REPO_ROOT/tools/microbench.py task > /tmp/prof_task.log 2>&1
tail -3 /tmp/prof_task.log
f=$(find /tmp/prof_task -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cp $f $GRAFT_REPO_ROOT/gpurun_out/task_kernel_stats.csv
head -12 $f
