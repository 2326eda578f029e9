"""Where does the GPU wait for the host?  Region timeline of the training step without a tracing profiler.

    python tools/timeline.py [--batch 32] [--steps 6]

HIP events are recorded at the region boundaries of the step (backbone fwd | head fwd + loss | head bwd | backbone
bwd | gradient finalize + Adam) in two modes:
  normal   the step exactly as bench.py runs it;
  primed   a long spin kernel is queued first, so the host has enqueued the WHOLE step before the GPU starts: region
           times are then pure GPU time.
normal - primed per region = GPU idle caused by the host in that region.  Host enqueue time per region is printed too.
Diagnostic only (a tracing profiler adds 10-20 us of host time per launch and moves the gaps around)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--sync-debug", action="store_true")
    args = ap.parse_args()
    import bench
    from grit_amd.amp import Bf16Compute
    from grit_amd.config import default_config
    from grit_amd.data import synthetic_batch
    from grit_amd.engine.caption_engine import build_optimizers
    bench._enable_tuned_gemms()
    device = torch.device("cuda", 0)
    config = default_config()
    model = bench.build(device, config).train()
    wrapped = Bf16Compute(model, bucket_mb=64)
    opts = build_optimizers(wrapped, config, mode="xe")
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    batches = [synthetic_batch(args.batch, 640, 640, 20, device=device, seed=i) for i in range(2)]

    marks = {}

    def mark(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks[name] = (ev, time.perf_counter())

    backbone = model.detector.backbone
    orig_forward = backbone.forward

    def backbone_forward(x):
        outs = orig_forward(x)
        mark("backbone_fwd_end")
        hooked = []
        pending = [len([o for o in outs if o.requires_grad])]

        def grad_arrived(g):
            pending[0] -= 1
            if pending[0] == 0:
                mark("head_bwd_end")
            return g
        for o in outs:
            if o.requires_grad:
                o.register_hook(grad_arrived)
            hooked.append(o)
        return hooked
    backbone.forward = backbone_forward

    def step(i, primed):
        marks.clear()
        b = batches[i % 2]
        torch.cuda.synchronize()
        if primed:
            torch.cuda._sleep(primed)
        mark("start")
        out = wrapped(b['samples'], b['captions'])
        opts['model'].zero_grad(set_to_none=False)
        opts['backbone'].zero_grad(set_to_none=False)
        target = b['captions'][:, 1:].contiguous()
        out = out[:, :-1].contiguous()
        loss = loss_fn(out.view(-1, out.shape[-1]).float(), target.view(-1))
        mark("head_fwd_end")
        loss.backward()
        mark("bwd_end")
        wrapped.finish_gradient_sync()
        opts['model'].step()
        opts['backbone'].step()
        wrapped.after_optimizer_step()
        mark("opt_end")
        torch.cuda.synchronize()
        order = ["start", "backbone_fwd_end", "head_fwd_end", "head_bwd_end", "bwd_end", "opt_end"]
        gpu = [marks[a][0].elapsed_time(marks[b_][0]) for a, b_ in zip(order, order[1:])]
        host = [(marks[b_][1] - marks[a][1]) * 1e3 for a, b_ in zip(order, order[1:])]
        return gpu, host

    for i in range(3):
        step(i, 0)
    if args.sync_debug:  # report every implicitly synchronizing torch call of one step (stack traces to stderr)
        torch.cuda.set_sync_debug_mode("warn")
        from grit_amd.engine.caption_engine import train_xe_step
        train_xe_step(wrapped, batches[0], opts, loss_fn)
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
    # calibrate the spin kernel to ~250 ms
    t0 = time.perf_counter(); torch.cuda._sleep(10_000_000); torch.cuda.synchronize(); per = (time.perf_counter() - t0) / 10_000_000
    spin = int(0.25 / per)
    names = ["backbone fwd", "head fwd+loss", "head bwd", "backbone bwd", "finalize+Adam"]
    res = {}
    for mode, primed in (("normal", 0), ("primed", spin)):
        acc_g, acc_h = [0.0] * 5, [0.0] * 5
        for i in range(args.steps):
            g, h = step(i, primed)
            acc_g = [a + b for a, b in zip(acc_g, g)]
            acc_h = [a + b for a, b in zip(acc_h, h)]
        res[mode] = ([a / args.steps for a in acc_g], [a / args.steps for a in acc_h])
    print(f"{'region':16s} {'GPU normal':>11s} {'GPU primed':>11s} {'idle':>8s} {'host enqueue':>13s}   [ms]")
    for k, n in enumerate(names):
        gn, gp, h = res["normal"][0][k], res["primed"][0][k], res["normal"][1][k]
        print(f"{n:16s} {gn:11.2f} {gp:11.2f} {gn - gp:8.2f} {h:13.2f}")
    print(f"{'total':16s} {sum(res['normal'][0]):11.2f} {sum(res['primed'][0]):11.2f} "
          f"{sum(res['normal'][0]) - sum(res['primed'][0]):8.2f} {sum(res['normal'][1]):13.2f}")
    print("note: in 'normal' mode each step starts from an idle GPU (synchronize before start), so the host has no lead from the\n"
          "previous step's tail; bench.py's steady state lets the host run ahead across step boundaries.")


if __name__ == "__main__":
    main()
