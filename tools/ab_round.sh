# Same-box A/B of library variants (tools/build_variant.sh): the bench line of every variant, alternating, N rounds.
#   TAG=r03k VARIANTS="base foo bar" ROUNDS=2 bash tools/ab_round.sh      (through gpurun; `base` = the in-tree library)
# Timing only: a variant may compute wrong values (ablations), so nothing is checked here.
R=$GRAFT_REPO_ROOT; TAG=${TAG:-ab}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in $(seq 1 ${ROUNDS:-2}); do for V in ${VARIANTS:-base}; do
  L=""; [ $V != base ] && L=$R/tools/ab/libnerf_$V.so
  NERF_HIP_LIB=$L timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline ${BENCH_EXTRA:---no-render} > $O/bench_${V}_$i.json 2> $O/bench_${V}_$i.err || exit 4
  python - <<PY
import json
d = json.loads(open('$O/bench_${V}_$i.json').read().strip().splitlines()[-1])
k = {x['name']: x['avg_launch_ms'] for x in d['kernels']}
print('$V', round(d['ms_per_step'], 4), ' '.join('%s %.4f' % (n.replace('nerf_', ''), k[n]) for n in ('nerf_density_fwd', 'nerf_field_fwd', 'nerf_field_bwd', 'nerf_table_bin_fill', 'nerf_table_update_binned') if n in k), (d.get('render') or {}).get('ms_per_image'))
PY
done; done
