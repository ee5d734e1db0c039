#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
NEREUS_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --no-cpu-baseline --developed 0 > $O/bench_slab1.json 2> $O/bench_slab1.err; echo "slab1 rc=$?" >> $O/progress.log
