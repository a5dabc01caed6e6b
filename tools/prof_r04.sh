# Round-4 profiles (run on the GPU box through gpurun): the DRIVER's bench command plain and under rocprofv3 (kernel
# trace + stats), its host / device timeline (hip trace + kernel trace), PMC traffic of the dominant kernel (one counter
# per pass, as MI355X_MICROARCH.md prescribes), the default 1024-step run, cfg 3 / cfg 5.  Everything lands in
# gpurun_out/r04_*; the summaries that are judged are copied to profiles/ by hand.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
run_trace() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/p_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $B "$@" > /tmp/p_$name.log 2>&1
  grep '^{"metric"' /tmp/p_$name.log | tail -1 > $O/r04_${name}_bench_line_under_rocprof.json
  cp $(find /tmp/p_$name -name "*kernel_stats.csv" | head -1) $O/r04_${name}_kernel_stats.csv
  cp $(find /tmp/p_$name -name "*kernel_trace.csv" | head -1) $O/r04_${name}_kernel_trace.csv
  echo "== $name"; cut -d, -f1-7 $O/r04_${name}_kernel_stats.csv | cut -c1-160 | head -7
}
run_pmc() {  # name, counter, bench args...
  name=$1; ctr=$2; shift; shift
  rm -rf /tmp/c_${name}_$ctr
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/c_${name}_$ctr -- python $B "$@" > /tmp/c_${name}_$ctr.log 2>&1
  cp $(find /tmp/c_${name}_$ctr -name "*counter_collection.csv" | head -1) $O/r04_${name}_pmc_${ctr}.csv
  echo "== pmc $name $ctr: $(wc -l < $O/r04_${name}_pmc_${ctr}.csv) rows"
}
timeout 600 python $B --gpus 1 --steps 20 --warmup 5 > $O/r04_driver_bench_line.json 2> /dev/null
run_trace driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
rm -rf /tmp/p_tl; timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/p_tl -- python $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /tmp/p_tl.log 2>&1
cp $(find /tmp/p_tl -name "*kernel_trace.csv" | head -1) $O/r04_timeline_kernel_trace.csv
cp $(find /tmp/p_tl -name "*hip_api_trace.csv" | head -1) $O/r04_timeline_hip_trace.csv
run_trace default --no-cpu-baseline --no-secondary
run_pmc driver WRITE_SIZE --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
run_pmc driver FETCH_SIZE --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
run_pmc default WRITE_SIZE --no-cpu-baseline --no-secondary --repeats 2
run_pmc default FETCH_SIZE --no-cpu-baseline --no-secondary --repeats 2
run_trace cfg3 --config cfg3 --no-cpu-baseline --steps 256 --warmup 32
run_trace cfg5 --config cfg5 --no-cpu-baseline --steps 256 --warmup 32
grep -h '"value"' $O/r04_cfg3_bench_line_under_rocprof.json $O/r04_cfg5_bench_line_under_rocprof.json | cut -c1-200

cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/r04_driver_pmc_WRITE_SIZE.csv $O/r04_driver_pmc_FETCH_SIZE.csv --kernel rate_kernel_gated --units-per-launch 81920 --out $O/r04_pmc_traffic_driver.json > /dev/null 2>&1 && grep -E '"(hbm_bytes_per_unit|traffic_over_algorithmic_kernel|grid_threads)"' $O/r04_pmc_traffic_driver.json
python tools/pmc_summary.py $O/r04_default_pmc_WRITE_SIZE.csv $O/r04_default_pmc_FETCH_SIZE.csv --kernel rate_kernel_gated --units-per-launch 4194304 --out $O/r04_pmc_traffic_default.json > /dev/null 2>&1 && grep -E '"(hbm_bytes_per_unit|traffic_over_algorithmic_kernel|grid_threads)"' $O/r04_pmc_traffic_default.json
RIAB_HIP_LIB=tools/exp/libpipe_prof.so timeout 300 python tools/pipe_profile.py 20 > $O/r04_pipe_profile_k20.txt 2>&1
timeout 300 python tools/slow_mode_probe.py 8 keep 2>&1 | grep -v amdgpu.ids > $O/r04_slow_mode_probe.txt
timeout 120 ./tools/queue_probe 16 keep 2>&1 > $O/r04_queue_probe_keep.txt
GPU_MAX_HW_QUEUES=8 timeout 120 ./tools/queue_probe 16 keep 2>&1 > $O/r04_queue_probe_keep_hwq8.txt
timeout 120 ./tools/wave_place > $O/r04_wave_place.txt 2>&1
head -3 $O/r04_driver_bench_line.json | cut -c1-300
