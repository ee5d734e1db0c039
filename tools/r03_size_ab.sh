#!/bin/bash
# packed vs scalar force walk at smaller scenes (broken-dam window of bench.py): where does the 5-wave packed kernel stop paying?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03size; mkdir -p $O
for cfg in 100,100,100 128,128,128 160,160,160; do
  for lib in main scalar; do
    if [ $lib = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$lib.so; fi
    timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --resting-steps 0 > $O/${cfg}_$lib.json 2> $O/${cfg}_$lib.err || { tail -3 $O/${cfg}_$lib.err; exit 1; }
    echo "$cfg $lib: $(python tools/bench_line.py $O/${cfg}_$lib.json | head -1)"
  done
done
