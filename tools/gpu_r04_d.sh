O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== default env, destroy"; timeout 120 ./tools/queue_probe 20 drop 2>&1 | tee $O/r04d_queue_drop.txt
echo "== GPU_MAX_HW_QUEUES=8, keep"; GPU_MAX_HW_QUEUES=8 timeout 120 ./tools/queue_probe 20 keep 2>&1 | tee $O/r04d_queue_keep8.txt
echo "== GPU_MAX_HW_QUEUES=8, destroy"; GPU_MAX_HW_QUEUES=8 timeout 120 ./tools/queue_probe 20 drop 2>&1 | tee $O/r04d_queue_drop8.txt
echo "== GPU_MAX_HW_QUEUES=8, default priority, keep"; GPU_MAX_HW_QUEUES=8 timeout 120 ./tools/queue_probe 20 keep 1 2>&1 | tee $O/r04d_queue_keep8_p1.txt
