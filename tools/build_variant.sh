#!/bin/bash
# Build a variant of libnerf_hip.so for same-box A/B timing (NERF_HIP_LIB=tools/ab/libnerf_<name>.so):
#   bash tools/build_variant.sh <name> "<-D defines>" <source stems to recompile with them...>
# The other objects come from the regular build (make -C nerf_amd/csrc first).
set -e
NAME=$1; DEFS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT/nerf_amd/csrc
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed"
mkdir -p /tmp/var_$NAME $ROOT/tools/ab
OBJS=""
for o in api_common render_kernels field_fwd field_bwd optim_kernels density_dx mlp_generic table_update sample_batch; do
  if [[ " $* " == *" $o "* ]]; then
    hipcc $F $DEFS -c $o.hip -o /tmp/var_$NAME/$o.o &
    OBJS="$OBJS /tmp/var_$NAME/$o.o"
  else
    OBJS="$OBJS build/$o.o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/libnerf_$NAME.so $OBJS
echo built tools/ab/libnerf_$NAME.so
