#!/bin/bash
# round 3, call M: the in-range divisions + two-record gathers on the in-tree build: exhaustive check of the forms, new guard test, parity +
# fuzz slice, A/B on the bench's state
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 600 tools/_bin/check_div2 > $O/check_div2.txt 2>&1 || { tail -30 $O/check_div2.txt; exit 1; }
tail -16 $O/check_div2.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_fast_arith_gpu.py -m gpu -q -x -k "not c3_iisph and not c5" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
bash tools/ab_flowing.sh "$@" > $O/ab.log 2>&1; tail -5 $O/ab.log
