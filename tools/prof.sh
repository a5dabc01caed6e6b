# The round's profiles (run on the GPU box through gpurun: `gpurun --timeout 900 -- 'bash tools/prof.sh r05'`): the DRIVER's
# bench command plain and under rocprofv3 (kernel trace + stats), PMC traffic of its dominant kernel (one counter per pass, as
# MI355X_MICROARCH.md prescribes; never together with a trace domain), the closed-loop forms (one kernel per step), cfg 3 /
# cfg 5; with a second argument `world`: only the one-world task's kernels, `driver`: only the driver's command.  Everything lands in gpurun_out/<round>_*; the summaries that are judged are copied to profiles/ by hand.
R=${1:-r06}
exec </dev/null
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
first() { find "$1" -name "$2" 2>/dev/null | head -1; }
run_trace() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/p_$name
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $B "$@" > /tmp/p_$name.log 2>&1
  grep '^{"metric"' /tmp/p_$name.log | tail -1 > $O/${R}_${name}_bench_line_under_rocprof.json
  f=$(first /tmp/p_$name "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $O/${R}_${name}_kernel_stats.csv
  f=$(first /tmp/p_$name "*kernel_trace.csv"); [ -n "$f" ] && cp "$f" $O/${R}_${name}_kernel_trace.csv
  echo "== $name"; [ -f $O/${R}_${name}_kernel_stats.csv ] && cut -d, -f1-7 $O/${R}_${name}_kernel_stats.csv | cut -c1-170 | sed -n 1,6p
}
run_pmc() {  # name, counter, bench args...
  name=$1; ctr=$2; shift; shift
  rm -rf /tmp/c_${name}_$ctr
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/c_${name}_$ctr -- python $B "$@" > /tmp/c_${name}_$ctr.log 2>&1
  f=$(first /tmp/c_${name}_$ctr "*counter_collection.csv"); [ -n "$f" ] && cp "$f" $O/${R}_${name}_pmc_${ctr}.csv
  echo "== pmc $name $ctr: $( [ -f $O/${R}_${name}_pmc_${ctr}.csv ] && wc -l < $O/${R}_${name}_pmc_${ctr}.csv ) rows"
}
if [ "$2" = "world" ]; then  # only the one-world task (`bash tools/prof.sh r05 world`): tools/task_world_time.py under the kernel trace
  rm -rf /tmp/p_world
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_world -- python $GRAFT_REPO_ROOT/tools/task_world_time.py > $O/${R}_task_world_time_under_rocprof.txt 2>&1
  f=$(first /tmp/p_world "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $O/${R}_task_world_kernel_stats.csv
  cut -d, -f1-7 $O/${R}_task_world_kernel_stats.csv | cut -c1-170 | sed -n 1,8p
  exit 0
fi
timeout 400 python $B --gpus 1 --steps 20 --warmup 5 > $O/${R}_driver_bench_line.json 2> /dev/null
run_trace driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run_pmc driver WRITE_SIZE --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
run_pmc driver FETCH_SIZE --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
if [ "$2" = "driver" ]; then  # only the driver's command (`bash tools/prof.sh r05 driver`): line, trace, PMC traffic
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py $O/${R}_driver_pmc_WRITE_SIZE.csv $O/${R}_driver_pmc_FETCH_SIZE.csv --kernel rate_kernel_gated --units-per-launch 81920 --out $O/${R}_pmc_traffic_driver.json > /dev/null 2>&1 && grep -E '"(hbm_bytes_per_unit|traffic_over_algorithmic_kernel|grid_threads)"' $O/${R}_pmc_traffic_driver.json
  exit 0
fi
run_trace plan --plan --steps 256 --warmup 32 --no-cpu-baseline --no-secondary
run_pmc plan WRITE_SIZE --plan --steps 256 --warmup 32 --no-cpu-baseline --no-secondary --repeats 3
run_pmc plan FETCH_SIZE --plan --steps 256 --warmup 32 --no-cpu-baseline --no-secondary --repeats 3
run_trace task --task --steps 256 --warmup 32 --no-cpu-baseline --no-secondary
run_pmc task WRITE_SIZE --task --steps 256 --warmup 32 --no-cpu-baseline --no-secondary --repeats 3
run_pmc task FETCH_SIZE --task --steps 256 --warmup 32 --no-cpu-baseline --no-secondary --repeats 3
run_trace cfg3 --config cfg3 --no-cpu-baseline --steps 256 --warmup 32
run_trace cfg5 --config cfg5 --no-cpu-baseline --steps 256 --warmup 32
# (round 6) the closed loops of cfg 3 / cfg 5 (one step1 kernel + the boundary vector cells per step), the one-world task
# (one step1_task_kernel per step), cfg 3 in the 64-wall room at 1024 steps (the trajectory kernel's wall loops)
run_trace cfg3_plan --config cfg3 --plan --no-cpu-baseline --no-secondary --steps 256 --warmup 32
run_trace cfg5_plan --config cfg5 --plan --no-cpu-baseline --no-secondary --steps 256 --warmup 32
run_trace task_world --task-world --no-cpu-baseline --no-secondary --steps 256 --warmup 32
run_trace cfg3_64w --config cfg3_64w --no-cpu-baseline --no-secondary --steps 1024 --warmup 32
run_trace cfg3_1024 --config cfg3 --no-cpu-baseline --no-secondary --steps 1024 --warmup 32
run_trace cfg5_1024 --config cfg5 --no-cpu-baseline --no-secondary --steps 1024 --warmup 32
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/${R}_driver_pmc_WRITE_SIZE.csv $O/${R}_driver_pmc_FETCH_SIZE.csv --kernel rate_kernel_gated --units-per-launch 81920 --out $O/${R}_pmc_traffic_driver.json > /dev/null 2>&1 && grep -E '"(hbm_bytes_per_unit|traffic_over_algorithmic_kernel|grid_threads)"' $O/${R}_pmc_traffic_driver.json
python tools/pmc_summary.py $O/${R}_plan_pmc_WRITE_SIZE.csv $O/${R}_plan_pmc_FETCH_SIZE.csv --kernel step1_kernel --units-per-launch 4096 --out $O/${R}_pmc_traffic_plan.json > /dev/null 2>&1 && grep -E '"(hbm_bytes_per_unit|traffic_over_algorithmic_kernel|grid_threads)"' $O/${R}_pmc_traffic_plan.json
python tools/pmc_summary.py $O/${R}_task_pmc_WRITE_SIZE.csv $O/${R}_task_pmc_FETCH_SIZE.csv --kernel step1_task_kernel --units-per-launch 4096 --out $O/${R}_pmc_traffic_task.json > /dev/null 2>&1 && grep -E '"(hbm_bytes_per_unit|traffic_over_algorithmic_kernel|grid_threads)"' $O/${R}_pmc_traffic_task.json
echo "== done"
