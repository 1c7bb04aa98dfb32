# r03c probes: update pass with 4096-float partitions (no quarter re-scan, plain fill), update pass with no items,
# forward gathers with the big dense levels in brick order (timing only).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
timeout -k 10 120 python tools/table_update_bench.py > $O/tub_base.txt 2>&1 || exit 1; tail -1 $O/tub_base.txt
NERF_HIP_LIB=$R/tools/ab/libnerf_p4096.so timeout -k 10 120 python tools/table_update_bench.py > $O/tub_p4096.txt 2>&1 || exit 2; tail -1 $O/tub_p4096.txt
timeout -k 10 120 python tools/upd_noitems_probe.py > $O/noitems.txt 2>&1 || exit 3; tail -1 $O/noitems.txt
NERF_HIP_LIB=$R/tools/ab/libnerf_p4096.so timeout -k 10 120 python tools/upd_noitems_probe.py > $O/noitems_p4096.txt 2>&1 || exit 3; tail -1 $O/noitems_p4096.txt
for V in base brick; do
  L=""; [ $V = brick ] && L=$R/tools/ab/libnerf_brick.so
  NERF_HIP_LIB=$L timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$V.json 2> $O/bench_$V.err || exit 4
  python - <<PY
import json
d = json.loads(open('$O/bench_$V.json').read().strip().splitlines()[-1])
k = {x['name']: x['avg_launch_ms'] for x in d['kernels']}
print('$V', round(d['ms_per_step'], 4), 'density_fwd', k.get('nerf_density_fwd'), 'field_fwd', k.get('nerf_field_fwd'), 'fill', k.get('nerf_table_bin_fill'), 'render ms', round(d['render']['ms_per_image'], 2))
PY
done
