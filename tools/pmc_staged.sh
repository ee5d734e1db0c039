#!/bin/bash
# PMC passes over the gather kernels of one density+forces pass (tools/ablate_density.py, 2.1M particles). GPU box only.
# usage: bash tools/pmc_staged.sh <tag> [fast]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_${1:-x}
export NEREUS_ABLATE_FAST=${2:-0}
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/tools/ablate_density.py 128,128,128 > $OUT/$name.log 2>&1
  echo "$name done" >> $OUT/progress.log
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
