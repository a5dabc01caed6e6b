# -DRIAB_PIPE_PROFILE build of the library (per-row device time stamps of the flag-coupled pipeline): tools/exp/libpipe_prof.so
set -e
R=$(cd $(dirname $0)/.. && pwd); W=/tmp/ppobjs; mkdir -p $W $R/tools/exp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DRIAB_PIPE_PROFILE -I $R/include -I $R/ratinabox_amd/csrc"
for u in riab_agent riab_rates riab_bvc riab_ff riab_ovc riab_plan riab_task riab_env riab_simulate; do
  hipcc $F -c $R/ratinabox_amd/csrc/$u.hip -o $W/$u.o &
done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/libpipe_prof.so $W/*.o
ls -la $R/tools/exp/libpipe_prof.so
