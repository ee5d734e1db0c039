#!/bin/bash
# round 3, GPU call E: suite on the build with the deferred pack totals, diagnosis of fuzz seed 20728, oracle fuzz on the NaN seeds, slab soaks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 120 python tools/fuzz_step_diag.py 20728 4 > $O/diag_20728.txt 2>&1; cat $O/diag_20728.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout -k 10 300 python tools/fuzz_parity.py 200 30000 oracle > $O/fuzz_oracle.txt 2>&1; tail -3 $O/fuzz_oracle.txt
timeout -k 10 600 python tools/fuzz_slab.py 100 9900 > $O/fuzz_slab.txt 2>&1; tail -2 $O/fuzz_slab.txt
FUZZ_SLAB_BIG=1 timeout -k 10 600 python tools/fuzz_slab.py 30 9950 > $O/fuzz_slab_big.txt 2>&1; tail -2 $O/fuzz_slab_big.txt
timeout -k 10 300 python tools/fuzz_slab.py 15 9980 iisph > $O/fuzz_slab_iisph.txt 2>&1; tail -2 $O/fuzz_slab_iisph.txt
NEREUS_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/slab1.json 2> $O/slab1.err || { tail -3 $O/slab1.err; exit 1; }
python tools/bench_line.py $O/slab1.json
