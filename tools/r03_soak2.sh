#!/bin/bash
# mid-round soak on the build with the two-record gathers and the in-range divisions: production vs reference order, vs the oracle, slab path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03soak2; mkdir -p $O
timeout -k 10 700 python tools/fuzz_parity.py ${1:-1500} 70000 > $O/fuzz_parity.txt 2>&1; tail -1 $O/fuzz_parity.txt
timeout -k 10 300 python tools/fuzz_parity.py ${2:-300} 80000 oracle > $O/fuzz_oracle.txt 2>&1; tail -1 $O/fuzz_oracle.txt
timeout -k 10 300 python tools/fuzz_slab.py ${3:-80} 13000 > $O/fuzz_slab.txt 2>&1; tail -1 $O/fuzz_slab.txt
