#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
export NEREUS_ABLATE_NOREF=1
for pad in 0 16384 32768 65536; do NEREUS_DBG_LDS_PAD_S=$pad timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/occupancy.log 2>&1; echo "pad $pad" >> $O/occupancy.log; done
NEREUS_ABLATE_FAST=1 timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/occupancy.log 2>&1
NEREUS_STAGED=0 timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/occupancy.log 2>&1
echo "occ done" >> $O/progress.log
timeout -k 10 500 bash tools/pmc_staged.sh r02c_exact 0; echo "pmc exact rc=$?" >> $O/progress.log
timeout -k 10 500 bash tools/pmc_staged.sh r02c_fast 1; echo "pmc fast rc=$?" >> $O/progress.log
