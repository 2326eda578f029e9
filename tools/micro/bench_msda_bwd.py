"""MSDeformAttn backward at the benchmark's shapes (B = 32, 8 500 cells, 150 queries, 8 heads, 4 x 4 points, bf16 maps): the three
ways of forming the value gradient, us per call (HIP events over a loop; inputs L2/Infinity-Cache warm), on two point
distributions: `spread` and `cluster` (a freshly initialised decoder: every query's reference point near
the image centre, the four points of a level in one cell).   python tools/micro/bench_msda_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd.ops import msda as M  # noqa: E402


def inputs(kind, B=32, Lq=150):
    g = torch.Generator().manual_seed(0)
    shapes = torch.tensor([[80, 80], [40, 40], [20, 20], [10, 10]])
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = torch.randn(B, S, 8, 64, generator=g).bfloat16()
    if kind == "spread":
        ref = torch.rand(B, Lq, 1, 1, 1, 2, generator=g)
        loc = (ref + 0.05 * torch.randn(B, Lq, 8, 4, 4, 2, generator=g)).clamp(-0.05, 1.05)
    else:
        ref = 0.5 + 0.06 * torch.randn(B, Lq, 1, 1, 1, 2, generator=g)
        loc = ref + 0.002 * torch.randn(B, Lq, 8, 4, 4, 2, generator=g)
    aw = torch.softmax(torch.randn(B, Lq, 8, 16, generator=g), -1).view(B, Lq, 8, 4, 4)
    cot = torch.randn(B, Lq, 512, generator=g).bfloat16()
    return [t.cuda() for t in (value, shapes, lsi, loc, aw, cot)]


def timed(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    for kind in ("spread", "cluster"):
        value, shapes, lsi, loc, aw, cot = inputs(kind)
        line = [kind]
        ref = None
        methods = ((True, "sorted"), (True, "staged"), (False, "packed bf16"))
        if os.environ.get("ONLY_SORTED") == "1":
            methods = methods[:1]
        for acc, method in methods:
            M.F32_ACCUMULATE, M.F32_METHOD = acc, (method if acc else "sorted")
            gv, gl, ga = M.ms_deform_attn_backward(value, shapes, lsi, loc, aw, cot)
            if ref is None:
                ref = gv.float()
            err = float(torch.linalg.norm(gv.float() - ref) / torch.linalg.norm(ref))
            us = timed(lambda: M.ms_deform_attn_backward(value, shapes, lsi, loc, aw, cot))
            line.append("%s %.0f us (rel. diff to sorted %.1e)" % (method, us, err))
        print("   ".join(line), flush=True)


if __name__ == "__main__":
    main()
