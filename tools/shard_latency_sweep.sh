#!/bin/bash
# Sensitivity of the level-sharded step to the one figure of the wire model nobody can know before an 8-GPU run, the
# latency of a collective, with and without the ray all-gather prefetch (NERF_PREFETCH_RAYS; NerfRenderer.prefetch_batch).
#   TAG=r03e bash tools/shard_latency_sweep.sh        -> gpurun_out/$TAG/shard_latency_sensitivity.txt
set -u
TAG=${TAG:-lat}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render > $OUT/single.json 2> $OUT/single.err || exit 1
SINGLE=$(python -c "import json;print(json.load(open('$OUT/single.json'))['ms_per_step'])")
F=$OUT/shard_latency_sensitivity.txt
echo "# tools/shard_step_emulate.py --latency-us L (per collective) under the wire model, 20 steps, default two ray groups; single GPU $SINGLE ms on this box; pf1 / pf0 = ray all-gather of the next batch prefetched before the optimizer step / not" > $F
for W in ${WORLDS:-8 2}; do
  for L in ${LATENCIES:-20 60 120}; do
    for PF in 1 0; do
      NERF_PREFETCH_RAYS=$PF timeout -k 10 300 python tools/shard_step_emulate.py --world $W --steps 20 --latency-us $L --single-ms $SINGLE \
          > $OUT/lat_w${W}_l${L}_pf${PF}.log 2> $OUT/lat_w${W}_l${L}_pf${PF}.err || exit 1
      echo "w${W}_l${L}_pf${PF} $(tail -1 $OUT/lat_w${W}_l${L}_pf${PF}.log)" | tee -a $F
    done
  done
done
