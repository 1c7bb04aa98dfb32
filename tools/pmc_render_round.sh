# SQ / TCP / TCC counters of the forward kernels on the inference path (tools/render_profile.py), one pass per block.
#   TAG=r02n_pmc bash tools/pmc_render_round.sh        (NERF_COHERENT_RAYS=0 in the environment for the old traversal)
R=$GRAFT_REPO_ROOT; TAG=${TAG:-render_pmc}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TRAIN_STEPS=0
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_VALU"
B="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum"
C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
timeout -s KILL 120 rocprofv3 --pmc $A --output-format csv -d $O/a -o a -- python $R/tools/render_profile.py 1 > $O/a.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc $B --output-format csv -d $O/b -o b -- python $R/tools/render_profile.py 1 > $O/b.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/c -o c -- python $R/tools/render_profile.py 1 > $O/c.log 2>&1 || echo "pass C failed (counter names?)"
cd $R && python tools/pmc_kernel.py "field_fwd_kernel" $O/a $O/b $O/c > $O/counters.txt
find $O -name "*counter_collection.csv" -delete
cat $O/counters.txt
