#!/bin/bash
# soak of the build with the batched IISPH list walks and the wall workgroups in the list kernels: general fuzz (every fifth seed IISPH), IISPH slabs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03soak4; mkdir -p $O
timeout -k 10 700 python tools/fuzz_parity.py ${1:-1500} 110000 > $O/fuzz_parity.txt 2>&1; tail -1 $O/fuzz_parity.txt
timeout -k 10 300 python tools/fuzz_parity.py ${2:-300} 120000 oracle > $O/fuzz_oracle.txt 2>&1; tail -1 $O/fuzz_oracle.txt
timeout -k 10 400 python tools/fuzz_slab.py ${3:-40} 15000 iisph > $O/fuzz_slab_iisph.txt 2>&1; tail -1 $O/fuzz_slab_iisph.txt
timeout -k 10 300 python tools/fuzz_long.py 60 1100 > $O/fuzz_long.txt 2>&1; tail -1 $O/fuzz_long.txt
