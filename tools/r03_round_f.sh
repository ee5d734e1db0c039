#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 120 python tools/fuzz_dens_diag.py 30171 > $O/diag_30171.txt 2>&1; cat $O/diag_30171.txt
timeout -k 10 120 python tools/fuzz_step_diag.py 20728 4 > $O/diag_20728.txt 2>&1; cat $O/diag_20728.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
NEREUS_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/slab1.json 2> $O/slab1.err || { tail -3 $O/slab1.err; exit 1; }
python tools/bench_line.py $O/slab1.json
timeout -k 10 300 python bench.py --no-cpu-baseline --resting-steps 0 > $O/single.json 2> $O/single.err || { tail -3 $O/single.err; exit 1; }
python tools/bench_line.py $O/single.json
