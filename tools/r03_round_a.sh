#!/bin/bash
# round 3, GPU call A: full GPU suite, the default bench line, the flowing-window profile, busy counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python tools/bench_line.py $O/bench_default.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -5 $O/bench_driver.err; exit 1; }
python tools/bench_line.py $O/bench_driver.json
bash tools/profile_flowing.sh r03 || exit 1
cd $GRAFT_REPO_ROOT && bash tools/busy_counters.sh > $O/busy.txt 2>&1 || { tail -5 $O/busy.txt; exit 1; }
cat $GRAFT_REPO_ROOT/$O/busy.txt
