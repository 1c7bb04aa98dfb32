# SQ stall / instruction-mix counters of the field kernels: three rocprofv3 --pmc passes (<= 8 SQ counters each).
#   TAG=r02c bash tools/pmc_sq_round.sh
R=$GRAFT_REPO_ROOT; TAG=${TAG:-sq}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM"
C="SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"
timeout -s KILL 120 rocprofv3 --pmc $A --output-format csv -d $O/sq_a -o a -- python $R/bench.py --steps 2 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-render > $O/sq_a.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc $B --output-format csv -d $O/sq_b -o b -- python $R/bench.py --steps 2 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-render > $O/sq_b.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/sq_c -o c -- python $R/bench.py --steps 2 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-render > $O/sq_c.log 2>&1 || exit 5
cd $R && python tools/pmc_kernel.py "${KERNELS:-field_bwd_kernel|field_fwd_kernel|bin_fill_sorted|table_update_kernel}" $O/sq_a $O/sq_b $O/sq_c > $O/sq_counters.txt
find $O -name "*counter_collection.csv" -delete
cat $O/sq_counters.txt
