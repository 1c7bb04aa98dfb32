#!/bin/bash
# Builds libnerf_hip variants with parts of the fill pass switched off (NERF_FILL_ABL, timing only: results are wrong) into
# tools/ab/ (not tracked); run on the GPU box: for A in 0 1 2 3 4; do NERF_HIP_LIB=tools/ab/lib_abl$A.so python tools/table_update_bench.py; done
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
mkdir -p ../../tools/ab
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed"
for A in ${ABLS:-0 1 2 3 4}; do
  hipcc $FLAGS -DNERF_FILL_ABL=$A -c table_update.hip -o /tmp/table_update_abl$A.o
  OBJS=$(ls build/*.o | grep -v table_update.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_abl$A.so $OBJS /tmp/table_update_abl$A.o
done
