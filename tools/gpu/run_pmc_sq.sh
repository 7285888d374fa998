# manual helper: SQ / LDS / TA counters of the kernels of one plan.  usage: run_pmc_sq.sh TAG SIZE VIEWS BATCH K
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02/sq_$1
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/a -- python tools/gpu/gpu_one_plan.py $2 $3 $4 $5 > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/b -- python tools/gpu/gpu_one_plan.py $2 $3 $4 $5 > $O/b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/c -- python tools/gpu/gpu_one_plan.py $2 $3 $4 $5 > $O/c.log 2>&1
echo done
