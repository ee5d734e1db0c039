#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
for v in "" dw6 dw7; do
  lib=""; [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/tools/_bin/libnereus_hip_$v.so
  echo "variant=${v:-default(unbounded)}" >> $O/bench.log
  NEREUS_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --developed 0 --steps 60 --warmup 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f'%d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_warmup_avg'].items()})" >> $O/bench.log
done
echo "variants done" >> $O/progress.log
