#!/bin/bash
# kernel trace of the slab path on one GPU (world = 1): usage (GPU box): bash tools/profile_slab.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$R/gpurun_out/prof_slab_$TAG
mkdir -p $OUT
export NEREUS_BENCH_FORCE_SLAB=1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace done" >> $OUT/progress.log
