#!/bin/bash
# PMC passes over the density/forces kernels (ablate tool, 2.1M particles). Run on the GPU box via gpurun.
# rocprofv3 needs the program itself after "--" (no wrappers) and counters in their own passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_${1:-x}
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/tools/ablate_density.py 128,128,128 > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum

run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum

ls $OUT/*/ > /dev/null
