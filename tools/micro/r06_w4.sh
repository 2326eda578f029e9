#!/bin/bash
# round 6: 224-row tiles + residual epilogue of the four-wave GEMM -- parity tests, then the timing table (same box)
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q > $O/w4_tests.log 2>&1; tail -n 15 $O/w4_tests.log
timeout 900 python tools/micro/bench_w4_vs_lib.py > $O/w4_vs_lib${TAG}.txt 2>&1; tail -n 21 $O/w4_vs_lib${TAG}.txt
