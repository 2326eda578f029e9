R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python - <<'PY' 2>&1 | grep -v Warn | tail -30
import os, sys, torch
sys.path.insert(0, '.')
import bench
from grit_amd.ops import gemm as G
orig = G.linear_bias_gelu
calls = []
def spy(x2, w, b, row_scale=None, rows_per_sample=0):
    calls.append((tuple(x2.shape), None if row_scale is None else (row_scale.dtype, int((row_scale == 0).sum()), row_scale.numel(), rows_per_sample)))
    return orig(x2, w, b, row_scale, rows_per_sample)
G.linear_bias_gelu = spy
import grit_amd.ops.mlp as M
M.G.linear_bias_gelu = spy
from grit_amd.amp import Bf16Compute
from grit_amd.config import default_config
from grit_amd.data import synthetic_batch
from grit_amd.engine.caption_engine import build_optimizers, train_xe_step
dev = torch.device('cuda', 0)
cfg = default_config()
model = bench.build(dev, cfg).train()
wrapped = Bf16Compute(model, bucket_mb=64)
opts = build_optimizers(wrapped, cfg, mode='xe')
batch = synthetic_batch(32, 640, 640, 20, device=dev, seed=0)
train_xe_step(wrapped, batch, opts, torch.nn.NLLLoss(ignore_index=1))
torch.cuda.synchronize()
for c in calls: print(c)
PY
