#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
export NEREUS_ABLATE_NOREF=1
for v in "" w6 w5 b2 b2w6; do
  lib=""; [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/tools/_bin/libnereus_hip_$v.so
  for f in 0 1; do
    echo "variant=${v:-default} fast=$f" >> $O/ablate.log
    NEREUS_HIP_LIB=$lib NEREUS_ABLATE_FAST=$f timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/ablate.log 2>&1
  done
done
echo "ablate done" >> $O/progress.log
cat > /tmp/pmc2.sh <<'EOS'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
export NEREUS_ABLATE_FAST=$2
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --pmc "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/tools/ablate_density.py 128,128,128 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LEVEL_WAVES
EOS
timeout -k 10 400 bash /tmp/pmc2.sh r02e_exact 0; echo "pmc exact rc=$?" >> $O/progress.log
timeout -k 10 400 bash /tmp/pmc2.sh r02e_fast 1; echo "pmc fast rc=$?" >> $O/progress.log
