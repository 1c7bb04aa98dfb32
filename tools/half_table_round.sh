R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_data_parallel.py -m gpu -q -k half --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
case $rc in 124|137|134|139) exit $rc;; esac
for P in float32 float16; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --table-precision $P > $O/bench_$P.json 2> $O/bench_$P.err || exit 4
  python - <<PY
import json
d = json.loads(open('$O/bench_$P.json').read().strip().splitlines()[-1])
print('$P', round(d['ms_per_step'], 4), d['render']['ms_per_image'], [(k['name'], k['ms_per_step']) for k in d['kernels'][:12]])
PY
done
