# r03g: field backward with the dW products on split-bf16 MFMAs (default build) against the f32-input MFMAs (tools/ab/libnerf_dwf32.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_properties.py tests/test_gpu_guard.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
case $rc in 124|137|134|139) exit $rc;; esac
for i in 1 2; do for V in base nofix; do
  L=""; [ $V != base ] && L=$R/tools/ab/libnerf_$V.so
  NERF_HIP_LIB=$L timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-render > $O/bench_${V}_$i.json 2> $O/bench_${V}_$i.err || exit 4
  python - <<PY
import json
d = json.loads(open('$O/bench_${V}_$i.json').read().strip().splitlines()[-1])
k = {x['name']: x['avg_launch_ms'] for x in d['kernels']}
print('$V', round(d['ms_per_step'], 4), 'field_bwd', k.get('nerf_field_bwd'), 'fill', k.get('nerf_table_bin_fill'), 'update', k.get('nerf_table_update_binned'))
PY
done; done
