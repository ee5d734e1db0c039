#!/bin/bash
# config C3 (IISPH, 4.1 M particles) on the in-tree library and on tools/_bin variants; with --parity <variant> the IISPH GPU tests run on it first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c3ab; mkdir -p $O
if [ "$1" = "--parity" ]; then
  NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$2.so timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "iisph" > $O/pytest_$2.log 2>&1 || { tail -15 $O/pytest_$2.log; exit 1; }
  tail -1 $O/pytest_$2.log; shift; shift
fi
for v in main "$@"; do
  if [ $v = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$v.so; fi
  timeout -k 10 300 python bench.py --solver iisph --config C3 --no-cpu-baseline > $O/c3_$v.json 2> $O/c3_$v.err || { tail -3 $O/c3_$v.err; exit 1; }
  python - "$O/c3_$v.json" "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s C3 ms/step %.4f"%(sys.argv[2], d["ms_per_step"]), {k:round(v,4) for k,v in d["stage_ms_warmup_avg"].items()})
PY
done
