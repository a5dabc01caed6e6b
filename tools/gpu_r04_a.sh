# Round 4, first GPU session: placement facts, the 20-step region variant by variant, the row-by-row pipeline profile,
# the whole GPU test suite, the driver's command.  Everything lands in gpurun_out/r04a_*.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== wave_place"; timeout 120 ./tools/wave_place > $O/r04a_wave_place.txt 2>&1; cat $O/r04a_wave_place.txt
echo "== probe"; timeout 900 python tools/r04_probe.py 20 300 > $O/r04a_probe.txt 2>&1; tail -25 $O/r04a_probe.txt
echo "== pipe profile (reserved / always)"
for g in reserved always; do
  RIAB_GATE=$g RIAB_HIP_LIB=tools/exp/libpipe_prof.so timeout 300 python tools/pipe_profile.py 20 > $O/r04a_pipe_profile_$g.txt 2>&1
  head -30 $O/r04a_pipe_profile_$g.txt
done
RIAB_GATE=reserved RIAB_PUB_SINGLE_ROWS=8 RIAB_POLL_SLEEP=16 RIAB_HIP_LIB=tools/exp/libpipe_prof.so timeout 300 python tools/pipe_profile.py 20 > $O/r04a_pipe_profile_reserved_8_16.txt 2>&1
head -30 $O/r04a_pipe_profile_reserved_8_16.txt
echo "== driver line"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04a_driver_bench_line.json 2> $O/r04a_driver_bench_err.txt; cut -c1-600 $O/r04a_driver_bench_line.json; tail -3 $O/r04a_driver_bench_err.txt
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/r04a_gpu_tests.txt; cat $O/r04a_gpu_tests.txt
