#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
export NEREUS_ABLATE_NOREF=1
timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/ablate.log 2>&1
NEREUS_ABLATE_FAST=1 timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/ablate.log 2>&1
NEREUS_STAGED=0 timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/ablate.log 2>&1
echo "ablate done" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_ns_exact.json 2> $O/bench_ns_exact.err; echo "ns exact rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline --arith fast > $O/bench_ns_fast.json 2> $O/bench_ns_fast.err; echo "ns fast rc=$?" >> $O/progress.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
