# One profiling round on the GPU box: plain bench, kernel trace, HBM traffic passes, MFMA counter pass.
#   TAG=r02a bash tools/profile_round.sh        (run through gpurun; results under gpurun_out/$TAG)
# Every rocprofv3 --pmc pass is a run of its own with the program directly after `--` (no shell in between).
set -o pipefail
R=$GRAFT_REPO_ROOT; TAG=${TAG:-r02a}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/bench_under_rocprof.json 2> $O/prof.err || exit 2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python $R/bench.py --steps 2 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-render > $O/pmc_f.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python $R/bench.py --steps 2 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-render > $O/pmc_w.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- python $R/bench.py --steps 2 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-render > $O/pmc_m.log 2>&1 || exit 5
cd $R && python tools/pmc_summary.py --steps 4 --build $TAG --out $O/traffic.json $O/pmc_f $O/pmc_w > $O/pmc_traffic.txt
cd $R/tools && python mfma_summary.py --build $TAG --out $O/mfma.json $O/pmc_m > $O/mfma.txt; cd $R
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# the per-ray kernels at 65 536 rays (where their bandwidth, not launch latency, is visible), under the kernel trace
cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rays -o rays -- python $R/tools/per_ray_bench.py > $O/per_ray.json 2> $O/per_ray.err || exit 6
cd $R; find $O/prof_rays -name "*kernel_stats.csv" -exec cp {} $O/per_ray_kernel_stats.csv \;
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline'],d['cpu_baseline'],d['render'])"
head -8 $O/kernel_stats.csv; head -14 $O/pmc_traffic.txt; cat $O/mfma.txt
