#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
if grep -q "Memory access fault" $O/pytest.log; then echo "FAULT - stopping" >> $O/progress.log; exit 1; fi
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_ns.json 2> $O/bench_ns.err; echo "ns rc=$?" >> $O/progress.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config C5 --precision 64 --kernel-set monaghan --developed 0 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/progress.log
NEREUS_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_slab1.json 2> $O/bench_slab1.err; echo "slab1 rc=$?" >> $O/progress.log
timeout -k 10 500 bash tools/profile_developed.sh r02_dev; echo "prof dev rc=$?" >> $O/progress.log
timeout -k 10 400 bash tools/pmc_staged.sh r02_final 0; echo "pmc rc=$?" >> $O/progress.log
