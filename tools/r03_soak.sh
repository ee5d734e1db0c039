#!/bin/bash
# end-of-round soak on the final build (packed force walk): production vs reference order, vs the oracle, slab path, long runs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03soak; mkdir -p $O
timeout -k 10 1000 python tools/fuzz_parity.py 3200 50000 > $O/fuzz_parity.txt 2>&1; tail -1 $O/fuzz_parity.txt
timeout -k 10 500 python tools/fuzz_parity.py 500 60000 oracle > $O/fuzz_oracle.txt 2>&1; tail -1 $O/fuzz_oracle.txt
FUZZ_SLAB_BIG=1 timeout -k 10 500 python tools/fuzz_slab.py 60 12000 > $O/fuzz_slab_big.txt 2>&1; tail -1 $O/fuzz_slab_big.txt
timeout -k 10 400 python tools/fuzz_slab.py 150 12100 > $O/fuzz_slab.txt 2>&1; tail -1 $O/fuzz_slab.txt
timeout -k 10 500 python tools/fuzz_long.py 150 700 > $O/fuzz_long.txt 2>&1; tail -1 $O/fuzz_long.txt
