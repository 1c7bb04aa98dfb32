# r03f: update pass "sum the bin, then stream p, m, v" against the shipped "request p, m, v, then sum" (tools/ab/libnerf_upd1.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
for i in 1 2; do
timeout -k 10 120 python tools/table_update_bench.py > $O/tub_base_$i.txt 2>&1 || exit 1; tail -1 $O/tub_base_$i.txt
NERF_HIP_LIB=$R/tools/ab/libnerf_upd1.so timeout -k 10 120 python tools/table_update_bench.py > $O/tub_upd1_$i.txt 2>&1 || exit 2; tail -1 $O/tub_upd1_$i.txt
done
