#!/bin/bash
# round 3, GPU call D: suite on the final build, 4-rank rehearsal of the N-rank bench (gloo, one GPU), randomised soaks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
NEREUS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 4 --config 64,56,56 --steps 20 --warmup 5 > $O/bench_4rank.json 2> $O/bench_4rank.err || { tail -15 $O/bench_4rank.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03e/bench_4rank.json").read().strip().splitlines()[-1])
print("4 ranks:", d["ms_per_step"], d["value"], d["config"]["spin_up_steps"], d["cfl_ok"], d["developed"], d["config"]["sort"], d["config"]["particles"])
PY
NEREUS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29579 bench.py --gpus 3 --solver iisph --config 48,40,40 --steps 10 --warmup 3 --iisph-max-iters 3 > $O/bench_3rank_iisph.json 2> $O/bench_3rank_iisph.err || { tail -15 $O/bench_3rank_iisph.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03e/bench_3rank_iisph.json").read().strip().splitlines()[-1])
print("3 ranks iisph:", d["ms_per_step"], d["value"], {k:v for k,v in d["config"].items() if "iisph" in k}, d["config"]["workload"][-80:])
PY
timeout -k 10 900 python tools/fuzz_parity.py 1300 20000 > $O/fuzz_parity.txt 2>&1; tail -2 $O/fuzz_parity.txt
timeout -k 10 600 python tools/fuzz_parity.py 400 30000 oracle > $O/fuzz_oracle.txt 2>&1; tail -2 $O/fuzz_oracle.txt
timeout -k 10 600 python tools/fuzz_slab.py 120 9000 > $O/fuzz_slab.txt 2>&1; tail -2 $O/fuzz_slab.txt
FUZZ_SLAB_BIG=1 timeout -k 10 600 python tools/fuzz_slab.py 40 9500 > $O/fuzz_slab_big.txt 2>&1; tail -2 $O/fuzz_slab_big.txt
timeout -k 10 300 python tools/fuzz_slab.py 20 9700 iisph > $O/fuzz_slab_iisph.txt 2>&1; tail -2 $O/fuzz_slab_iisph.txt
timeout -k 10 600 python tools/fuzz_long.py 60 500 > $O/fuzz_long.txt 2>&1; tail -2 $O/fuzz_long.txt
