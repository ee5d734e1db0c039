#!/bin/bash
# occupancy sensitivity of the density kernel (dynamic LDS padding lowers the workgroups per CU): fixed-state timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/occ; mkdir -p $O; : > $O/log.txt
# (round 3: the padding is a compile-time constant, -DNRS_DBG_LDS_PAD=bytes — the library reads no environment variables;
#  build the variants HERE before the gpurun call: for p in 0 5000 13000 25000; do tools/build_variant.sh pad$p -DNRS_DBG_LDS_PAD=$p; done)
NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_pad0.so timeout -k 10 300 python tools/density_ablate2.py save 760 /tmp/dev760.npz >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
for pad in 0 5000 13000 25000; do
  export NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_pad$pad.so
  echo "pad $pad" >> $O/log.txt
  timeout -k 10 120 python tools/density_ablate2.py time rest >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
  timeout -k 10 120 python tools/density_ablate2.py time /tmp/dev760.npz >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
done
grep -v Warning $O/log.txt
