#!/bin/bash
# A/B of environment-selected variants of the level-sharded step under the modelled wire, one world:
#   TAG=r02i_var W=2 VARIANTS="NERF_SHARD_RANGED_FILL=0 A=1:B=2"  (several variables of one variant joined with :) bash tools/shard_variants.sh
set -u
TAG=${TAG:-variants}
W=${W:-2}
SINGLE=${SINGLE:-4.13}
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for V in "BASE=1" ${VARIANTS:-}; do
  i=$((i+1))
  echo "== $V"
  env ${V//:/ } timeout -k 10 300 python tools/shard_step_emulate.py --world $W --steps 20 --single-ms $SINGLE --latency-us ${LAT:-20} \
      > $OUT/w${W}_v${i}.log 2> $OUT/w${W}_v${i}.err || exit 1
  tail -1 $OUT/w${W}_v${i}.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['max_ms'], d['mean_ms'], d.get('projected_weak_scaling_efficiency'))"
done
