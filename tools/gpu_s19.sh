cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s19.log 2>&1; echo rc $?; grep -v "^$" gpurun_out/s19.log | head -60 | cut -c1-300
