#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout 120 tools/micro/bin/atomic_rate > gpurun_out/r03/atomic_rate.txt 2>&1; cat gpurun_out/r03/atomic_rate.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -15 gpurun_out/r03/gpu_tests.log
