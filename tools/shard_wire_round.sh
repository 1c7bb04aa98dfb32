#!/bin/bash
# Level-sharded step of every rank of a pretend world with the modelled xGMI wire (tools/shard_step_emulate.py), for
# worlds 2 / 4 / 8 and the ray-group counts given in RAYGROUPS, next to the single-GPU step measured on the same box.
#   TAG=r02c RAYGROUPS="1 4" bash tools/shard_wire_round.sh
set -u
TAG=${TAG:-wire}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render > $OUT/single.json 2> $OUT/single.err || exit 1
SINGLE=$(python -c "import json;print(json.load(open('$OUT/single.json'))['ms_per_step'])")
echo "single-GPU ms/step: $SINGLE"
for W in ${WORLDS:-2 4 8}; do
  NERF_SHARD_GROUPS=1 timeout -k 10 300 python tools/shard_step_emulate.py --world $W --steps 20 --link-gbs 0 \
      > $OUT/shard_nowire_w${W}.log 2> $OUT/shard_nowire_w${W}.err || exit 1
  tail -1 $OUT/shard_nowire_w${W}.log
  for G in ${RAYGROUPS:-1 4}; do
    NERF_SHARD_GROUPS=$G timeout -k 10 300 python tools/shard_step_emulate.py --world $W --steps 20 --single-ms $SINGLE \
        > $OUT/shard_wire_w${W}_g${G}.log 2> $OUT/shard_wire_w${W}_g${G}.err || exit 1
    tail -1 $OUT/shard_wire_w${W}_g${G}.log
  done
done
