#!/bin/bash
# round 3, GPU call B: new tests, the matrix-core tile-cost experiment, the 2-rank slab bench rehearsal at full size (gloo, one GPU)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -k "nrs_step_returns or refshim or slab_rccl or bench_line or phase_guards or iteration" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout -k 10 120 tools/_bin/mfma_tile_cost > $O/mfma_tile_cost.txt 2>&1; cat $O/mfma_tile_cost.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err || { tail -5 $O/bench_driver.err; exit 1; }
python tools/bench_line.py $O/bench_driver.json
NEREUS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank.json 2> $O/bench_2rank.err || { tail -15 $O/bench_2rank.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03c/bench_2rank.json").read().strip().splitlines()[-1])
print("2 ranks:", d["ms_per_step"], d["value"], d["config"]["spin_up_steps"], d["cfl_ok"], d["developed"], d["config"]["sort"])
PY
