#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
export NEREUS_ABLATE_NOREF=1
run() { echo "$1" >> $O/ablate.log; shift; env "$@" timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/ablate.log 2>&1; }
run "staged exact"  NEREUS_ABLATE_FAST=0
run "staged fast"   NEREUS_ABLATE_FAST=1
run "old exact"     NEREUS_STAGED=0 NEREUS_ABLATE_FAST=0
run "staged exact nobound" NEREUS_ABLATE_NOBOUND=1 NEREUS_ABLATE_FAST=0
run "staged fast nobound"  NEREUS_ABLATE_NOBOUND=1 NEREUS_ABLATE_FAST=1
echo "ablate done" >> $O/progress.log
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_ns_exact.json 2> $O/bench_ns_exact.err; echo "ns exact rc=$?" >> $O/progress.log
NEREUS_STAGED=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_ns_old.json 2> $O/bench_ns_old.err; echo "ns old rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline --arith fast > $O/bench_ns_fast.json 2> $O/bench_ns_fast.err; echo "ns fast rc=$?" >> $O/progress.log
