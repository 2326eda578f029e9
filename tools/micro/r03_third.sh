#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_stream_kernels_gpu.py -q -m gpu -k "dropout_mask" 2>&1 | tail -5
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r03/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r03/gpu_tests.log | tail -15
