cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lin320
mkdir -p $O
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o out --output-format csv -- python3 $R/tools/one_gemm.py lin 320 320 72 128 > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS && \
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE && \
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum && \
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum && \
run tcc2 TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum
find $O -name "*counter_collection.csv" | head
