#!/bin/bash
# Copy the summaries of one tools/profile_round.sh run from gpurun_out/<tag> (scratch) into profiles/ (tracked).
#   bash tools/save_profiles.sh r03d
T=$1; O=gpurun_out/$T; P=profiles
cp $O/bench.json $P/${T}_bench.json
cp $O/bench_under_rocprof.json $P/${T}_bench_under_rocprof.json
cp $O/kernel_stats.csv $P/${T}_bench_kernel_stats.csv
cp $O/pmc_traffic.txt $P/${T}_pmc_traffic.txt
cp $O/mfma.txt $P/${T}_mfma.txt
cp $O/traffic.json $P/traffic.json
cp $O/mfma.json $P/mfma.json
[ -f $O/per_ray.json ] && cp $O/per_ray.json $P/${T}_per_ray_65536.json && cp $O/per_ray_kernel_stats.csv $P/${T}_per_ray_65536_kernel_stats.csv
[ -f $O/pytest.log ] && tail -3 $O/pytest.log > $P/${T}_gpu_tests.txt
ls $P | grep "^$T"
