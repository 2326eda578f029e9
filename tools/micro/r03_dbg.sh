#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_stream_kernels_gpu.py -q -m gpu 2>&1 | tail -2
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from grit_amd.ops.linear import column_sum
x = torch.randn(51200, 1536, device='cuda').bfloat16()
for _ in range(3): column_sum(x, torch.bfloat16)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): column_sum(x, torch.bfloat16)
b.record(); torch.cuda.synchronize()
print("column_sum [51200, 1536] bf16: %.1f us (colsum + slab sum)" % (a.elapsed_time(b) / 30 * 1e3))
PY
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-analysis --steps 40 --warmup 15 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; done
