#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
T="tests/test_slab_gloo.py::test_slab_hip_engine_coherent_resort"
run() { echo "$1" >> $O/bisect.log; shift; env "$@" timeout -k 10 200 python -m pytest $T -m gpu -q 2>&1 | tail -2 >> $O/bisect.log; }
run "default" A=1
run "WALL_PASS=0" NEREUS_WALL_PASS=0
run "WALL_PASS=0 PCT=12" NEREUS_WALL_PASS=0 NEREUS_RESORT_MAX_PCT=12
run "WALL_PASS=0 INPLACE=0" NEREUS_WALL_PASS=0 NEREUS_SLAB_INPLACE=0
echo "bisect done" >> $O/progress.log
export NEREUS_ABLATE_NOREF=1
abl() { echo "$1" >> $O/ablate.log; shift; env "$@" timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/ablate.log 2>&1; }
abl "old+wallpass" A=1
abl "old no wallpass" NEREUS_WALL_PASS=0
abl "fast" NEREUS_ABLATE_FAST=1
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_ns_exact.json 2> $O/bench_ns_exact.err; echo "ns exact rc=$?" >> $O/progress.log
