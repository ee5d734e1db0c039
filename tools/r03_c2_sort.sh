cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c2
for args in "" "--full-sort"; do
  timeout -k 10 120 python bench.py --config C2 --no-cpu-baseline $args > gpurun_out/r03c2/c2$args.json 2> gpurun_out/r03c2/c2$args.err || { tail -3 gpurun_out/r03c2/c2$args.err; exit 1; }
  python tools/bench_line.py gpurun_out/r03c2/c2$args.json
done
for args in "" "--full-sort"; do
  timeout -k 10 120 python bench.py --config 160,160,160 --no-cpu-baseline --resting-steps 0 $args > gpurun_out/r03c2/m4$args.json 2> gpurun_out/r03c2/m4$args.err || { tail -3 gpurun_out/r03c2/m4$args.err; exit 1; }
  python tools/bench_line.py gpurun_out/r03c2/m4$args.json
done
