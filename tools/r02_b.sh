#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_ns_exact.json 2> $O/bench_ns_exact.err; echo "ns exact rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline --arith fast > $O/bench_ns_fast.json 2> $O/bench_ns_fast.err; echo "ns fast rc=$?" >> $O/progress.log
NEREUS_STAGED=0 timeout -k 10 300 python bench.py --no-cpu-baseline --developed 0 > $O/bench_ns_old.json 2> $O/bench_ns_old.err; echo "ns old rc=$?" >> $O/progress.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config C2 --arith fast --developed 0 > $O/bench_c2_fast.json 2> $O/bench_c2_fast.err; echo "c2 rc=$?" >> $O/progress.log
