O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_drv -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p_drv.log 2>&1
cp $(find /tmp/p_drv -name "*kernel_trace.csv" | head -1) $O/s21_driver_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_def -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/p_def.log 2>&1
cp $(find /tmp/p_def -name "*kernel_trace.csv" | head -1) $O/s21_default_kernel_trace.csv
