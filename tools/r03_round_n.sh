#!/bin/bash
# round 3, call N: parity of the packed density walk (in-tree build), then the fused bench A/B against the variants named on the command line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "not c3_iisph and not c5 and not velocity_bar" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
bash tools/r03_bench_ab.sh "$@"
