#!/bin/bash
# rocprofv3 view of one BASELINE configuration's bench line (kernel trace + HBM counters in their own passes), summaries to gpurun_out/prof_<tag>/:
#   bash tools/profile_config.sh <tag> <name> <particles> <keep last N steps> "<workload text>" <bench.py args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; NAME=$2; NP=$3; KEEP=$4; WL=$5; shift 5
OUT=/tmp/prof_$TAG; mkdir -p $OUT $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/bench.py --no-cpu-baseline "$@" --steps 5 > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/bench.py --no-cpu-baseline "$@" --steps 5 > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
cd $R && NEREUS_PROFILE_OUT=$R/gpurun_out/prof_$TAG python3 tools/summarize_profile.py $OUT $NAME "$WL" $NP $KEEP > $R/gpurun_out/prof_$TAG/summary.txt 2>&1 || { tail -5 $R/gpurun_out/prof_$TAG/summary.txt; exit 1; }
grep "^{" $OUT/trace.log | tail -1 > $R/gpurun_out/prof_$TAG/bench_line_under_profiler.json
