#!/bin/bash
# end-of-round soak on the final build: production vs reference order, vs the oracle, slab path, long runs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03soak3; mkdir -p $O
timeout -k 10 1000 python tools/fuzz_parity.py 2000 90000 > $O/fuzz_parity.txt 2>&1; tail -1 $O/fuzz_parity.txt
timeout -k 10 500 python tools/fuzz_parity.py 300 95000 oracle > $O/fuzz_oracle.txt 2>&1; tail -1 $O/fuzz_oracle.txt
FUZZ_SLAB_BIG=1 timeout -k 10 500 python tools/fuzz_slab.py 40 14000 > $O/fuzz_slab_big.txt 2>&1; tail -1 $O/fuzz_slab_big.txt
timeout -k 10 400 python tools/fuzz_slab.py 100 14100 > $O/fuzz_slab.txt 2>&1; tail -1 $O/fuzz_slab.txt
timeout -k 10 500 python tools/fuzz_long.py 120 900 > $O/fuzz_long.txt 2>&1; tail -1 $O/fuzz_long.txt
