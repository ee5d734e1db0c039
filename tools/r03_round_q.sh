#!/bin/bash
# round 3, call Q: batched list walks of the IISPH chain on the in-tree build: parity (SESPH + IISPH + fuzz + refshim), then C3 against variants
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_refshim.py tests/test_slab_gloo.py -m gpu -q -x -k "not velocity_bar" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
bash tools/r03_c3_ab.sh "$@"
