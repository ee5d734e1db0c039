cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abf
timeout -k 10 300 tools/_bin/check_div2 > gpurun_out/abf/check_div2.txt 2>&1; cat gpurun_out/abf/check_div2.txt
bash tools/ab_flowing.sh --parity packed5d packed5 packed5d packed4d
