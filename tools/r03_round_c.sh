#!/bin/bash
# round 3, GPU call C: full suite on the renamed build, tile-cost experiment at two hit rates, one bench line per config
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 120 tools/_bin/mfma_tile_cost 262144 768 > $O/mfma_tile_cost_12pct.txt 2>&1; cat $O/mfma_tile_cost_12pct.txt
timeout -k 10 120 tools/_bin/mfma_tile_cost 262144 256 > $O/mfma_tile_cost_45pct.txt 2>&1; cat $O/mfma_tile_cost_45pct.txt
bash tools/bench_configs.sh
