set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python $R/bench.py --steps 2 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-render > $O/pmc_f.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python $R/bench.py --steps 2 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-render > $O/pmc_w.log 2>&1 || exit 4
cd $R && python tools/pmc_summary.py --steps 4 --out $O/traffic.json $O/pmc_f $O/pmc_w > $O/pmc_traffic.txt
find $O -name "*counter_collection.csv" -delete
head -6 $O/pmc_traffic.txt
