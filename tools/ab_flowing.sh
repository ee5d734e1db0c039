#!/bin/bash
# round 3 A/B on the bench's own state: save the NS scene after 3000 steps at dt = 2.5e-4 s once, then time the density and force stages
# of `main` and of every tools/_bin variant named on the command line at rest and on that state; with --parity <variant> the GPU parity
# tests run on that variant first (production == reference order bit for bit is what guards an arithmetic rewrite)
cd $GRAFT_REPO_ROOT
O=gpurun_out/abf; mkdir -p $O; : > $O/log.txt
if [ "$1" = "--parity" ]; then
  NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$2.so timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "not velocity_bar and not c3_iisph and not c5 and not slab" > $O/pytest_$2.log 2>&1 || { tail -15 $O/pytest_$2.log; exit 1; }
  tail -2 $O/pytest_$2.log; shift; shift
fi
NEREUS_ABL_DT=2.5e-4 timeout -k 10 300 python tools/density_ablate2.py save 3000 /tmp/flow3000.npz >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
for v in main "$@"; do
  if [ $v = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$v.so; fi
  timeout -k 10 120 python tools/density_ablate2.py time rest >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
  timeout -k 10 120 python tools/density_ablate2.py time /tmp/flow3000.npz >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
done
grep -v Warning $O/log.txt
