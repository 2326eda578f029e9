#!/bin/bash
# A/B of GRIT_WGRAD_STREAM (weight gradients on a second stream) on one box: tests first, then the bench line twice each.
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_gemm_gpu.py tests/test_stream_kernels_gpu.py -x -q -m gpu 2>&1 | tail -5
for k in 0 1 0 1; do
  GRIT_WGRAD_STREAM=$k timeout 400 python bench.py --no-cpu-baseline --no-analysis --steps 40 --warmup 15 2>/dev/null | tail -1 \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('WGRAD_STREAM=$k', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', 'loss', d.get('final_loss'))"
done | tee gpurun_out/r02/ab_wgrad_stream.txt
