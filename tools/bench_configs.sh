#!/bin/bash
# one bench line per BASELINE config that fits one GPU (+ the slab path on one rank)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03cfg; mkdir -p $O
timeout -k 10 200 python bench.py --config C2 --no-cpu-baseline > $O/c2.json 2> $O/c2.err || { tail -3 $O/c2.err; exit 1; }
timeout -k 10 300 python bench.py --solver iisph --config C3 --no-cpu-baseline > $O/c3.json 2> $O/c3.err || { tail -3 $O/c3.err; exit 1; }
timeout -k 10 300 python bench.py --precision 64 --kernel-set monaghan --config C5 --no-cpu-baseline > $O/c5.json 2> $O/c5.err || { tail -3 $O/c5.err; exit 1; }
NEREUS_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/slab1.json 2> $O/slab1.err || { tail -3 $O/slab1.err; exit 1; }
python - <<'PY'
import json
for n in ("c2","c3","c5","slab1"):
    d=json.loads(open("gpurun_out/r03cfg/%s.json"%n).read().strip().splitlines()[-1])
    dev=d.get("developed") or {}; rest=d.get("resting") or {}; end=dev.get("at_end") or {}
    print(n, d["config"]["workload"][:60], "| ms/step", round(d["ms_per_step"],4), "value %.3e"%d["value"], "dtype", d["dtype"], "cfl_ok", d.get("cfl_ok"),
          "spin-up", d["config"].get("spin_up_steps"), "| resting", round(rest.get("ms_per_step",0),4), "| neighbours", end.get("neighbours_mean"), "overflow", end.get("hit_list_overflow_fraction"),
          "vmax", dev.get("vmax"), "whole-step frac", round(d["roofline"]["whole_step"].get("frac", d["roofline"]["whole_step"].get("frac_per_gpu", 0)),4))
PY
