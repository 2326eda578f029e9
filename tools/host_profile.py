"""Host-side cost of enqueuing one training step (cProfile of the main thread; the autograd thread is not included)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from grit_amd.amp import Bf16Compute
    from grit_amd.config import default_config
    from grit_amd.data import synthetic_batch
    from grit_amd.engine.caption_engine import build_optimizers, train_xe_step
    bench._enable_tuned_gemms()
    device = torch.device("cuda", 0)
    config = default_config()
    model = bench.build(device, config).train()
    wrapped = Bf16Compute(model, bucket_mb=64)
    opts = build_optimizers(wrapped, config, mode="xe")
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    batch = synthetic_batch(32, 640, 640, 20, device=device, seed=0)
    for _ in range(3):
        train_xe_step(wrapped, batch, opts, loss_fn)
    torch.cuda.synchronize()
    # forward-only and full-step host times without the profiler
    t0 = time.perf_counter(); out = wrapped(batch['samples'], batch['captions']); t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("forward enqueue %.1f ms" % ((t1 - t0) * 1e3))
    del out
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        train_xe_step(wrapped, batch, opts, loss_fn)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
