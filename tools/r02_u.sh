#!/bin/bash
# A/B of library variants on the NS scene: tools/r02_u.sh <outdir-tag> <variant names...> ("main" = the in-tree library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  if [ $v = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?" >> $O/progress.log
done
python - "$O" "$@" <<'PY'
import json,sys
O=sys.argv[1]
for n in sys.argv[2:]:
    try:
        d=json.loads(open("%s/bench_%s.json"%(O,n)).read().strip().splitlines()[-1])
        print("%-10s"%n, "ms/step", round(d["ms_per_step"],4), {k:round(v["ms"],4) for k,v in d["per_stage_roofline"].items()}, "| developed", round(d["developed"]["ms_per_step"],4), {k:round(v,4) for k,v in d["developed"]["stage_ms"].items()})
    except Exception as e: print(n, "failed", e)
PY
