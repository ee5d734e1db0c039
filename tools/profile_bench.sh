#!/bin/bash
# Round profile of the bench workload: kernel-trace stats, then HBM traffic counters in their own passes.
# usage (GPU box, via gpurun): bash tools/profile_bench.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
echo "trace done" >> $OUT/progress.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1
echo "fetch done" >> $OUT/progress.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/write.log 2>&1
echo "write done" >> $OUT/progress.log
