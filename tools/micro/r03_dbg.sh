#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
python tools/bench_small_wgrad.py 2>&1 | tail -12
bash tools/micro/prof_step.sh 2>&1 | grep -E "wgrad_small|slab_sum|colsum|GPU busy"
bash tools/micro/ab_env.sh GRIT_WGRAD_SMALL 0 1
