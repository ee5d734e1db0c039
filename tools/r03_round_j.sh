#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r03_round_a.sh || exit 1
cd $GRAFT_REPO_ROOT && bash tools/bench_configs.sh
