#!/bin/bash
# round-6 same-box baseline: GEMM tables the policy is read off + the driver's bench command (numbers land in gpurun_out/r06/)
O=gpurun_out/r06; mkdir -p $O
timeout 600 python tools/micro/bench_w4_vs_lib.py > $O/w4_vs_lib${TAG}.txt 2>&1
timeout 600 python tools/micro/bench_fused_variants.py > $O/fused_variants${TAG}.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver${TAG}.json 2> $O/bench_driver${TAG}.err
tail -n 30 $O/w4_vs_lib${TAG}.txt; cat $O/fused_variants${TAG}.txt; tail -c 1500 $O/bench_driver${TAG}.json
