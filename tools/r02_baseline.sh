#!/bin/bash
# round-2 baseline on the GPU box: suite, then one bench line per BASELINE config, then the re-sort crossover sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_ns.json 2> $O/bench_ns.err; echo "ns rc=$?" >> $O/progress.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config C2 --developed 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/progress.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config C5 --precision 64 --kernel-set monaghan --developed 0 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/progress.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config C3 --solver iisph --steps 30 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/progress.log
for f in 0.03 0.10 0.15 0.20 0.30 0.40 0.50 0.65 0.80; do timeout -k 10 120 tools/_bin/bench_coherent_sort 10077696 $f >> $O/sort_sweep.log 2>&1; done; echo "sweep done" >> $O/progress.log
