#!/bin/bash
# fixed-state timing of the density / force stages for library variants; stops at the first failing step
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O; : > $O/log.txt
if [ "$1" = "--parity" ]; then shift
  timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x > $O/pytest.log 2>&1 || { tail -15 $O/pytest.log; exit 1; }
  tail -2 $O/pytest.log
  if [ -n "$PARITY_LIB" ]; then
    NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$PARITY_LIB.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x > $O/pytest2.log 2>&1 || { tail -15 $O/pytest2.log; exit 1; }
    tail -2 $O/pytest2.log
  fi
fi
timeout -k 10 300 python tools/density_ablate2.py save 760 /tmp/dev760.npz >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
for v in main "$@"; do
  if [ $v = main ]; then unset NEREUS_HIP_LIB; else export NEREUS_HIP_LIB=$PWD/tools/_bin/libnereus_hip_$v.so; fi
  timeout -k 10 120 python tools/density_ablate2.py time rest >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
  timeout -k 10 120 python tools/density_ablate2.py time /tmp/dev760.npz >> $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
done
grep -v Warning $O/log.txt
