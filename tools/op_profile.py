"""Attribute GPU time of the training step to torch ops and to the source lines that issue them (torch.profiler).

    python tools/op_profile.py [--batch 32] [--steps 2] > gpurun_out/op_profile.txt

rocprofv3 tells which kernels are hot; this tells who launches the long tail of elementwise / copy / reduce kernels
(which module line), so they can be fused or removed at the source.  Diagnostic only."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--top", type=int, default=220)
    args = ap.parse_args()
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
    batches = [synthetic_batch(args.batch, 640, 640, 20, device=device, seed=i) for i in range(2)]
    for i in range(3):
        train_xe_step(wrapped, batches[i % 2], opts, loss_fn)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for i in range(args.steps):
            train_xe_step(wrapped, batches[i % 2], opts, loss_fn)
        torch.cuda.synchronize()

    def dev_time(e):
        return getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)

    by_site = collections.defaultdict(lambda: [0.0, 0])
    by_op = collections.defaultdict(lambda: [0.0, 0])
    for e in prof.events():
        t = dev_time(e)
        if not t or e.name.startswith(("void ", "Cijk", "_ZN", "Custom_Cijk", "__amd", "Memset", "Memcpy")):
            continue  # kernels are listed by rocprofv3 (tools/steady_profile.py); here only the ops that launch them
        site = "?"
        for fr in (e.stack or []):
            if "/grit_amd/" in fr or "/bench.py" in fr:
                site = fr.split("/grit_amd/")[-1] if "/grit_amd/" in fr else fr
                break
        shapes = str(e.input_shapes)[:70] if e.input_shapes else ""
        by_site[(e.name, site)][0] += t
        by_site[(e.name, site)][1] += 1
        by_op[(e.name, shapes)][0] += t
        by_op[(e.name, shapes)][1] += 1
    n = args.steps
    total = sum(v[0] for v in by_op.values())
    print(f"total device time {total / n / 1e3:.2f} ms/step over {n} steps\n\n== by (op, issuing line) ==")
    for (name, site), (t, c) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:args.top]:
        print(f"{t / n / 1e3:8.3f} ms/step  x{c / n:6.1f}  {name[:48]:48s}  {site[:90]}")
    print("\n== by (op, input shapes) ==")
    for (name, shapes), (t, c) in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:args.top]:
        print(f"{t / n / 1e3:8.3f} ms/step  x{c / n:6.1f}  {name[:48]:48s}  {shapes}")


if __name__ == "__main__":
    main()
