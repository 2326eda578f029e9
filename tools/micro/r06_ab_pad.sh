#!/bin/bash
# window attention backward: d(pad) through the qkv bias sum, one zero fill for all blocks -- tests, then A/B of both knobs together
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_winattn_gpu.py tests/test_drop_path_gpu.py tests/test_graph_step_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -4
for pass in 1 2 3; do for v in 1 0; do
GRIT_WINATTN_PAD_VIA_BIAS=$v GRIT_WINATTN_ZERO_ARENA=$v python bench.py --steps 30 --warmup 8 --no-analysis --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad_via_bias+arena=$v pass=$pass', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'img/s', 'loss', round(d['final_loss'],4))"
done; done | tee gpurun_out/r06/ab_pad_via_bias.txt
