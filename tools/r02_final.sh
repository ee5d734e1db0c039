#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
if grep -q "Memory access fault" $O/pytest.log; then echo "FAULT - stopping" >> $O/progress.log; exit 1; fi
timeout -k 10 400 python bench.py > $O/bench_ns.json 2> $O/bench_ns.err; echo "ns rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline --arith fast > $O/bench_ns_fast.json 2> $O/bench_ns_fast.err; echo "ns fast rc=$?" >> $O/progress.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config C2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/progress.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config C5 --precision 64 --kernel-set monaghan > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/progress.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config C3 --solver iisph --steps 30 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/progress.log
timeout -k 10 300 python tools/regime_probe.py NS 3000 250 > $O/regime_ns.jsonl 2> $O/regime_ns.err; echo "regime rc=$?" >> $O/progress.log
timeout -k 10 400 bash tools/profile_bench.sh r02_rest; echo "prof rest rc=$?" >> $O/progress.log
