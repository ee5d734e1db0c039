#!/bin/bash
# strip-major visiting order: parity with the map forced on, then bench A/B over the number of strips
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O; : > $O/log.txt
NEREUS_VISIT_STRIPS=4 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_slab_gloo.py -m gpu -q -x > $O/pytest.log 2>&1 || { tail -15 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for ns in 0 2 4 8 16; do
  NEREUS_VISIT_STRIPS=$ns timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_$ns.json 2> $O/bench_$ns.err || { tail -5 $O/bench_$ns.err; exit 1; }
done
python - <<'PY'
import json
for n in (0,2,4,8,16):
    d=json.loads(open("gpurun_out/r02z/bench_%d.json"%n).read().strip().splitlines()[-1])
    print("strips %2d"%n, "ms/step", round(d["ms_per_step"],4), {k:round(v["ms"],4) for k,v in d["per_stage_roofline"].items()}, "| developed", round(d["developed"]["ms_per_step"],4), {k:round(v,4) for k,v in d["developed"]["stage_ms"].items()})
PY
