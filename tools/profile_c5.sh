#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02b_c5; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --precision 64 --kernel-set monaghan --config C5 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/bench.py --precision 64 --kernel-set monaghan --config C5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/bench.py --precision 64 --kernel-set monaghan --config C5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/write.log 2>&1 || exit 1
cd $R
NEREUS_BENCH_FORCE_SLAB=1 NEREUS_BENCH_REBALANCE=20 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/slab_rebal.json 2> gpurun_out/slab_rebal.err || { tail -5 gpurun_out/slab_rebal.err; exit 1; }
tail -c 300 gpurun_out/slab_rebal.json
