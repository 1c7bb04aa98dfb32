#!/bin/bash
# Same-box A/B of environment variants of the single-GPU bench step: ms/step of each variant, alternating, REPS times.
#   TAG=ab VARIANTS="BASE=1 NERF_BWD_GROUPS=2" REPS=2 bash tools/ab_env.sh
set -u
TAG=${TAG:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for r in $(seq 1 ${REPS:-2}); do
  i=0
  for V in ${VARIANTS:-BASE=1}; do
    i=$((i+1))
    env ${V//,/ } timeout -k 10 200 python bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-render > $OUT/v${i}_r$r.json 2> $OUT/v${i}_r$r.err || exit 1
    python -c "import json;d=json.load(open('$OUT/v${i}_r$r.json'));print('$V', round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d['roofline_second']['avg_launch_ms'])"
  done
done
