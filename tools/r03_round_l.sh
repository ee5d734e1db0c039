#!/bin/bash
# final verification of round 3: build check on the box, smoke, the whole GPU suite, the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -5 $O/bench_driver.err; exit 1; }
python tools/bench_line.py $O/bench_driver.json
