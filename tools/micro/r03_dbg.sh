#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_stream_kernels_gpu.py tests/test_gemm_gpu.py -q -m gpu 2>&1 | tail -3
for v in 0 1; do echo "GRIT_SLAB_WIDE=$v"; GRIT_SLAB_WIDE=$v SHAPES=51200x2048x512,51200x512x512,51200x1536x512,12800x4096x1024 timeout 300 python tools/micro/bench_wgrad_tn.py 2>&1 | grep -v amdgpu | cut -c1-200; done
GRIT_AB_OUT=gpurun_out/r03/ab2 bash tools/micro/ab_env.sh GRIT_SLAB_WIDE 0 1
