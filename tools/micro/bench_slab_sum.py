"""grit_slab_sum_grouped on the shapes a Swin block's backward gives it: GB/s of the fp32 slices it reads (HIP events, rotating buffers
larger than the Infinity Cache so the slices come from HBM, and a second pass with ONE buffer = slices still in the Infinity Cache)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd.ops.linear import SlabGroup


def timed(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for name, jobs in (("mlp pair 2 x [8, 2048x512]", [(8, 2048 * 512), (8, 2048 * 512)]),
                   ("qkv + proj [16, 1536x512] + [16, 512x512]", [(16, 1536 * 512), (16, 512 * 512)]),
                   ("qkv alone [21, 1536x512]", [(21, 1536 * 512)]), ("proj alone [64, 512x512]", [(64, 512 * 512)])):
    for nbuf in (8, 1):
        bufs = [[torch.randn(1, S, n, device="cuda") for S, n in jobs] for _ in range(nbuf)]
        outs = [torch.empty(1, n, device="cuda", dtype=torch.bfloat16) for S, n in jobs]
        state = {"i": 0}

        def run():
            g = SlabGroup()
            for p, o in zip(bufs[state["i"] % nbuf], outs):
                g.add(p, torch.bfloat16, out=o)
            g.run()
            state["i"] += 1
        us = timed(run)
        mb = sum(S * n * 4 for S, n in jobs) / 1e6
        print("%-44s %s  %6.1f us  %6.1f MB  %5.2f TB/s (MB/us)" % (name, "HBM  " if nbuf > 1 else "cache", us, mb, mb / us))
