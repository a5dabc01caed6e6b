# whole-library builds with extra flags (timing experiments): ratinabox_amd/lib/variants/lib_<name>.so, loaded through
# RIAB_HIP_LIB.  Usage: tools/build_policy_variants.sh name:"-DFLAG ..." ...
set -e
R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/ratinabox_amd/lib/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I $R/include -I $R/ratinabox_amd/csrc"
UNITS="riab_rates riab_agent riab_bvc riab_ff riab_ovc riab_plan riab_task riab_task_world riab_env riab_simulate riab_step1"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}; W=/tmp/polobjs_$name; mkdir -p $W
  for u in $UNITS; do hipcc $F $flags -c $R/ratinabox_amd/csrc/$u.hip -o $W/$u.o & done
done; wait
for v in "$@"; do
  name=${v%%:*}; W=/tmp/polobjs_$name; OBJS=""; for u in $UNITS; do OBJS="$OBJS $W/$u.o"; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ratinabox_amd/lib/variants/lib_$name.so $OBJS
done
ls -la $R/ratinabox_amd/lib/variants/
