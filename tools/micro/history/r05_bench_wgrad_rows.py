"""grit_wgrad_tn_grouped with / without the drop-path factors at the Swin stage-2 shapes (two problems per launch, as the Mlp node and the
attention node launch them): us per launch, HIP events around loops of 20."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd import lib as _lib  # noqa: E402
from grit_amd.ops.linear import tn_slices  # noqa: E402

lib = _lib.load()
DEV = "cuda"


def run(problems, B, per, dropped, label):
    M = B * per
    scale = torch.full((B,), 1.2, device=DEV)
    for b in dropped:
        scale[b] = 0
    tiles = sum((N // 256) * (K // 256) for N, K in problems)
    want = max(1, 256 // tiles)
    keep = []
    for use in (False, True):
        table = (_lib.WgradJob * len(problems))()
        for t, (N, K) in enumerate(problems):
            dy = torch.randn(M, N, device=DEV).bfloat16()
            for b in dropped:
                dy[b * per:(b + 1) * per] = 0
            x = torch.randn(M, K, device=DEV).bfloat16()
            S = tn_slices(M, want)
            part = torch.empty((S, N, K), dtype=torch.float32, device=DEV)
            keep.append((dy, x, part))
            table[t] = _lib.WgradJob(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, N, K, S, part.data_ptr(), None,
                                     scale.data_ptr() if use else None, per if use else 0)
        for _ in range(3):
            assert lib.grit_wgrad_tn_grouped(table, len(problems), _lib.current_stream_ptr()) == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            lib.grit_wgrad_tn_grouped(table, len(problems), _lib.current_stream_ptr())
        b.record()
        torch.cuda.synchronize()
        print(f"{label}: factors {'on ' if use else 'off'} {a.elapsed_time(b) / 20 * 1e3:8.1f} us  ({len(dropped)} of {B} samples dropped, S = {S})", flush=True)


for dropped in ((), (3, 7, 12, 20, 29), tuple(range(0, 32, 3))):
    run([(512, 2048), (2048, 512)], 32, 1600, dropped, "stage 2 Mlp   (fc2 + fc1)")
    run([(512, 512), (1536, 512)], 32, 1600, dropped, "stage 2 attn  (proj + qkv)")
run([(256, 1024), (1024, 256)], 32, 6400, (1, 5, 9), "stage 1 Mlp")
