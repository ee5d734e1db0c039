#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
export NEREUS_ABLATE_NOREF=1
abl() { echo "$1" >> $O/ablate.log; shift; env "$@" timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/ablate.log 2>&1; }
abl "wallblocks" A=1
abl "no wallblocks" NEREUS_WALL_PASS=0
timeout -k 10 200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x > $O/pytest_parity.log 2>&1; echo "parity rc=$?" >> $O/progress.log
if grep -q "Memory access fault" $O/pytest_parity.log; then echo "FAULT - stopping" >> $O/progress.log; exit 1; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_ns_exact.json 2> $O/bench_ns_exact.err; echo "ns exact rc=$?" >> $O/progress.log
if grep -q "Memory access fault" $O/bench_ns_exact.err; then echo "FAULT - stopping" >> $O/progress.log; exit 1; fi
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_parity_gpu.py > $O/pytest_rest.log 2>&1; echo "rest rc=$?" >> $O/progress.log
