#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02y; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 40 --warmup 20 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline_gaps.py $f 20 > $OUT/gaps.txt 2>&1; cat $OUT/gaps.txt
