#!/bin/bash
# end-of-round-2 profiles: resting + developed kernel stats and HBM counters, gather-kernel PMC summary, busy fractions
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r02c_rest || exit 1
bash tools/profile_developed.sh r02c_dev || exit 1
bash tools/pmc_staged.sh r02c 0 || exit 1
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc2 && bash tools/busy_counters.sh > gpurun_out/pmc2/busy.txt 2>&1 || exit 1
cat gpurun_out/pmc2/busy.txt
