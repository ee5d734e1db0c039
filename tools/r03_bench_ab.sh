#!/bin/bash
# the driver's bench command on the in-tree library and on tools/_bin variants (timing A/B of the fused step): bash tools/r03_bench_ab.sh <variant>...
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
for v in main "$@"; do
  if [ $v = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$v.so; fi
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  echo "== $v"; python tools/bench_line.py $O/bench_$v.json
done
