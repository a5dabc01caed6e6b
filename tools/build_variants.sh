# ablation builds of the four-wave trajectory kernel (timing experiments): tools/exp/libt4_<bits>.so
set -e
R=$(cd $(dirname $0)/.. && pwd); W=/tmp/t4objs; mkdir -p $W $R/tools/exp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I $R/include -I $R/ratinabox_amd/csrc"
for u in riab_rates riab_bvc riab_ff riab_ovc riab_plan riab_task riab_env riab_simulate; do
  [ -f $W/$u.o ] && [ $W/$u.o -nt $R/ratinabox_amd/csrc/$u.hip ] || hipcc $F -c $R/ratinabox_amd/csrc/$u.hip -o $W/$u.o &
done; wait
for bits in "$@"; do
  if [ "$bits" = prof ]; then
  (hipcc $F -DRIAB_T4_PROFILE -c $R/ratinabox_amd/csrc/riab_agent.hip -o $W/agent_prof.o && hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/libt4_prof.so $W/agent_prof.o $W/riab_rates.o $W/riab_bvc.o $W/riab_ff.o $W/riab_ovc.o $W/riab_plan.o $W/riab_task.o $W/riab_env.o $W/riab_simulate.o) &
  continue; fi
  (hipcc $F -DRIAB_T4_ABLATE=$bits -c $R/ratinabox_amd/csrc/riab_agent.hip -o $W/agent_$bits.o && \
   hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/libt4_$bits.so $W/agent_$bits.o $W/riab_rates.o $W/riab_bvc.o $W/riab_ff.o $W/riab_ovc.o $W/riab_plan.o $W/riab_task.o $W/riab_env.o $W/riab_simulate.o) &
done; wait
ls -la $R/tools/exp/libt4_*.so
