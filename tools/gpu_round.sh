# One GPU session: the GPU test-suite, then a bench line.  A step that ends in a fault, an abort, a segmentation fault or
# a time limit ends the script there (nothing more is started on the GPU); an ordinary test failure does not.
#   TAG=r03a bash tools/gpu_round.sh [pytest args ...]
R=$GRAFT_REPO_ROOT; TAG=${TAG:-r03a}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 ${PYTEST_LIMIT:-700} python -u -m pytest ${@:-tests -m gpu} -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
case $rc in 124|137|134|139|-6|-11) exit $rc;; esac
if [ "${BENCH:-1}" = "1" ]; then
  timeout -k 10 ${BENCH_LIMIT:-300} python bench.py ${BENCH_ARGS:---no-cpu-baseline} > $O/bench.json 2> $O/bench.err
  rc=$?; echo "bench rc=$rc"; [ $rc -ne 0 ] && { tail -5 $O/bench.err; exit $rc; }
  python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['render'])
for r in d.get('per_ray_kernels_65536_rays') or []: print(r['name'], r['avg_launch_ms'], r['algorithmic_GBps'], r['hbm_frac'])
for k in d['kernels']: print(k['name'], k['ms_per_step'], k.get('hbm_frac'), k.get('mfma_f32_frac'))
PY
fi
