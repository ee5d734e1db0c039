#!/bin/bash
# round 3, GPU call G: soaks on the final build (vs the oracle incl. the NaN / inf seeds; production vs reference order), C4 strong-scaling rehearsal
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 600 python tools/fuzz_parity.py 400 30000 oracle > $O/fuzz_oracle.txt 2>&1; tail -2 $O/fuzz_oracle.txt
timeout -k 10 900 python tools/fuzz_parity.py 1500 40000 > $O/fuzz_parity.txt 2>&1; tail -2 $O/fuzz_parity.txt
NEREUS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29580 bench.py --gpus 2 --config C4 --scaling strong --steps 20 --warmup 5 > $O/bench_c4_strong_2rank.json 2> $O/bench_c4.err || { tail -15 $O/bench_c4.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03h/bench_c4_strong_2rank.json").read().strip().splitlines()[-1])
print("C4 strong, 2 ranks on one GPU:", d["ms_per_step"], d["value"], d["config"]["particles"], d["cfl_ok"], d["config"]["sort"], d["scaling"])
PY
timeout -k 10 300 python -m pytest tests -m gpu -q -k "fuzz or refshim or phase or surface" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
