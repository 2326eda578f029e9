#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
GRIT_TEST_SEED_GUARD=1 timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r03/gpu_tests_guard.log 2>&1; echo "gpu tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r03/gpu_tests_guard.log | tail -8
grep -E "^E  " gpurun_out/r03/gpu_tests_guard.log | cut -c1-300 | head -12
GRIT_AB_OUT=gpurun_out/r03/ab2 bash tools/micro/ab_env.sh GRIT_WGRAD_TN_PAIR 0 1
