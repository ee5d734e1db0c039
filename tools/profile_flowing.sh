#!/bin/bash
# Round-3 profile of the bench's own timed window (the broken dam: 3000 untimed spin-up steps at dt = 2.5e-4 s, then warm-up + timed
# steps): kernel trace over the default command (tools/summarize_profile.py ... 100 keeps the last 100 steps = the timed region),
# HBM counters in their own passes (last 5 dispatches kept).
# usage (GPU box, via gpurun): bash tools/profile_flowing.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
OUT=/tmp/prof_$TAG          # raw traces stay on the box (a 3120-step kernel trace is > 64 MiB); only the summaries travel back
mkdir -p $OUT $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
echo "trace done" >> $OUT/progress.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/bench.py --steps 5 --warmup 20 --resting-steps 0 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
echo "fetch done" >> $OUT/progress.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/bench.py --steps 5 --warmup 20 --resting-steps 0 --no-cpu-baseline "$@" > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
echo "write done" >> $OUT/progress.log
cd $R && NEREUS_PROFILE_OUT=$R/gpurun_out/prof_$TAG python3 tools/summarize_profile.py $OUT ${TAG}_ns10M_flowing "SESPH dam-break 216^3 = 10,077,696 particles + tank, fp32, Muller kernels, exact arithmetic; bench.py default command: 3000 untimed spin-up steps at dt = 2.5e-4 s, 20 warm-up, 100 timed steps (statistics over the last 100 steps = the timed region; HBM counters over the last 5 dispatches of 3025-step runs)" 10077696 100 > $R/gpurun_out/prof_$TAG/summary.txt 2>&1 || { tail -5 $R/gpurun_out/prof_$TAG/summary.txt; exit 1; }
cp $OUT/trace.log $R/gpurun_out/prof_$TAG/bench_line_under_profiler.json
