#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
export NEREUS_ABLATE_NOREF=1
for nb in "" 1; do for st in "" 0; do for f in 0 1; do
  echo "nobound=${nb:-0} staged=${st:-1} fast=$f" >> $O/ablate.log
  NEREUS_ABLATE_NOBOUND=$nb NEREUS_STAGED=$st NEREUS_ABLATE_FAST=$f timeout -k 10 120 python tools/ablate_density.py 128,128,128 >> $O/ablate.log 2>&1
done; done; done
echo "ablate done" >> $O/progress.log
timeout -k 10 900 python -m pytest tests/test_fast_arith_gpu.py tests/test_host_class.py tests/test_slab_gloo.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu-baseline --developed 2880 --developed-steps 100 --steps 20 --warmup 5 > $O/bench_ns_long.json 2> $O/bench_ns_long.err; echo "long rc=$?" >> $O/progress.log
