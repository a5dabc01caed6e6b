# timing builds of the one-launch step (csrc/riab_step1.hip): ratinabox_amd/lib/variants/libs1_<name>.so, loaded through
# RIAB_HIP_LIB.  Usage: tools/build_step1_variants.sh name:"-DFLAG ..." ...
set -e
R=$(cd $(dirname $0)/.. && pwd); W=/tmp/s1objs_$(echo $R | md5sum | cut -c1-8); mkdir -p $W $R/ratinabox_amd/lib/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I $R/include -I $R/ratinabox_amd/csrc"
NEWEST=$(ls -t $R/ratinabox_amd/csrc/*.h $R/include/riab_hip.h | head -1)   # (any header newer than an object: rebuild it)
UNITS="riab_rates riab_agent riab_bvc riab_ff riab_ovc riab_plan riab_task riab_task_world riab_env riab_simulate"
for u in $UNITS; do
  [ -f $W/$u.o ] && [ $W/$u.o -nt $R/ratinabox_amd/csrc/$u.hip ] && [ $W/$u.o -nt $NEWEST ] || hipcc $F -c $R/ratinabox_amd/csrc/$u.hip -o $W/$u.o &
done; wait
OBJS=""; for u in $UNITS; do OBJS="$OBJS $W/$u.o"; done
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  (hipcc $F $flags -c $R/ratinabox_amd/csrc/riab_step1.hip -o $W/step1_$name.o && \
   hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ratinabox_amd/lib/variants/libs1_$name.so $W/step1_$name.o $OBJS) &
done; wait
ls -la $R/ratinabox_amd/lib/variants/
