#!/bin/bash
# compact-scan bring-up: parity subset first, then A/B timing of the density stage (NS scene, resting + developed)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_fast_arith_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/progress.log
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_compact.json 2> $O/bench_compact.err; echo "compact rc=$?" >> $O/progress.log
NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_exactscan.so timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_exact.json 2> $O/bench_exact.err; echo "exact rc=$?" >> $O/progress.log
python - <<'PY'
import json
for n in ("compact","exact"):
    try:
        d=json.loads(open("gpurun_out/r02t/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, "ms/step", round(d["ms_per_step"],4), {k:round(v["ms"],4) for k,v in d["per_stage_roofline"].items()}, "developed", round(d["developed"]["ms_per_step"],4), {k:round(v,4) for k,v in d["developed"]["stage_ms"].items()}, "overflow", d["developed"]["hit_list_overflow_fraction"])
    except Exception as e: print(n, "failed", e)
PY
