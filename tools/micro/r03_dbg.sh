#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
GRIT_TEST_SEED_GUARD=1 timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_det_rows.py tests/test_stream_kernels_gpu.py tests/test_gemm_gpu.py tests/test_ddp_gloo.py -q -m gpu 2>&1 | tail -3
bash tools/micro/prof_step.sh > /dev/null 2>&1; head -1 gpurun_out/r03/steady.txt | cut -c1-150; grep -E "colsum|slab_sum|wgrad" gpurun_out/r03/steady.txt | cut -c1-140 | head -8
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-analysis --steps 40 --warmup 15 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; done
