#!/bin/bash
# The level-sharded step under the wire model with the fine pass's depths gathered group by group from inside the coarse pass
# (NERF_PIPELINED_DEPTH_GATHER=1, default) against ONE all-gather between the passes (=0); ray prefetch on in both.
#   TAG=r03t bash tools/shard_depth_gather_sweep.sh        -> gpurun_out/$TAG/shard_depth_gather.txt
set -u
TAG=${TAG:-dg}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render > $OUT/single.json 2> $OUT/single.err || exit 1
SINGLE=$(python -c "import json;print(json.load(open('$OUT/single.json'))['ms_per_step'])")
F=$OUT/shard_depth_gather.txt
echo "# tools/shard_step_emulate.py --latency-us L under the wire model, 20 steps, two ray groups, ray prefetch on; single GPU $SINGLE ms on this box; dg1 / dg0 = depth all-gather per ray group from inside the coarse pass / one all-gather between the passes" > $F
for W in ${WORLDS:-8 2}; do
  for L in ${LATENCIES:-20 60 120}; do
    for DG in 1 0; do
      NERF_PREFETCH_RAYS=1 NERF_PIPELINED_DEPTH_GATHER=$DG timeout -k 10 300 python tools/shard_step_emulate.py --world $W --steps 20 --latency-us $L --single-ms $SINGLE \
          > $OUT/dg_w${W}_l${L}_dg${DG}.log 2> $OUT/dg_w${W}_l${L}_dg${DG}.err || exit 1
      echo "w${W}_l${L}_dg${DG} $(tail -1 $OUT/dg_w${W}_l${L}_dg${DG}.log)" | tee -a $F
    done
  done
done
