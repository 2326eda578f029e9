"""Which torch ops launch the element-wise / copy kernels of a training step (torch.profiler, shapes + python stack)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from grit_amd.config import default_config
from grit_amd.data import synthetic_batch
from grit_amd.amp import Bf16Compute
from grit_amd.engine.caption_engine import build_optimizers, train_xe_step

device = torch.device("cuda", 0)
bench._enable_tuned_gemms()
config = default_config()
model = bench.build(device, config).train()
wrapped = Bf16Compute(model, bucket_mb=64, shard_optimizer=False)
optimizers = build_optimizers(wrapped, config, mode="xe")
loss_fn = torch.nn.NLLLoss(ignore_index=1)
batches = [synthetic_batch(32, 640, 640, 20, device=device, seed=i) for i in range(2)]
for i in range(4):
    train_xe_step(wrapped, batches[i % 2], optimizers, loss_fn)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train_xe_step(wrapped, batches[0], optimizers, loss_fn)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::addcmul", "aten::fill_", "aten::zero_", "aten::cat", "aten::clone", "aten::contiguous", "aten::_to_copy")
rows = {}
for e in prof.events():
    if e.name in want and e.device_time > 0:
        stack = [s for s in (e.stack or []) if "grit_amd" in s or "bench" in s][:3]
        key = (e.name, str(e.input_shapes)[:80], " <- ".join(s.split("/")[-1][:70] for s in stack))
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1; r[1] += e.device_time
tot = sum(r[1] for r in rows.values())
print("total device us in these ops: %.0f" % tot)
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%7.0f us  x%-3d %s %s  | %s" % (r[1], r[0], k[0], k[1], k[2]))
