#!/bin/bash
# PMC passes for ONE dc_gemm_conv shape (tools/one_gemm.py arguments are passed through).
# usage (on the GPU box): bash tools/pmc_one_gemm.sh <name> conv|lin|tconv|geglu <ci> <co> <H> <W>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
O=$R/gpurun_out/pmc_$name
mkdir -p $O
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o out --output-format csv -- python3 $R/tools/${PMC_SCRIPT:-one_gemm.py} $ARGS > $O/$n.log 2>&1 || { tail -5 $O/$n.log; return 1; }; }
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS && \
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC && \
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum && \
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
