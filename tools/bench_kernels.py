"""Developer micro-benchmarks of the individual HIP kernels (not the contract bench: see /bench.py).

    python tools/bench_kernels.py msda [--batch 8] [--iters 200]

Times launches with HIP events on torch's current stream and prints achieved algorithmic GB/s
(SURVEY 8d byte counts) next to the 8 TB/s HBM3E peak.
"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def config2(B, seed=0, Lq=150, device="cuda"):
    gen = torch.Generator().manual_seed(seed)
    shapes = torch.tensor([[80, 80], [40, 40], [20, 20], [10, 10]])
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, M, D, L, P = int(shapes.prod(1).sum()), 8, 64, 4, 4
    value = torch.randn(B, S, M, D, generator=gen)
    ref = torch.rand(B, Lq, 1, 1, 1, 2, generator=gen)
    loc = (ref + 0.05 * torch.randn(B, Lq, M, L, P, 2, generator=gen)).clamp(-0.05, 1.05)
    aw = torch.softmax(torch.randn(B, Lq, M, L * P, generator=gen), -1).view(B, Lq, M, L, P)
    return [t.to(device) for t in (value, shapes, lsi, loc, aw)]


def time_gpu(fn, iters, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def msda_bytes(B, bf16=False, S=8500, M=8, D=64, Lq=150, L=4, P=4):
    from grit_amd.ops.msda import _algorithmic_bytes
    sizes = (2, 4, 4) if bf16 else (4, 4, 4)  # value/out, locations+weights, grad_value
    return (_algorithmic_bytes("fwd", B, S, M, D, L, Lq, P, *sizes), _algorithmic_bytes("bwd", B, S, M, D, L, Lq, P, *sizes))


def bench_msda(args):
    from grit_amd.ops.msda import ms_deform_attn_forward, ms_deform_attn_backward
    res = []
    for B in args.batch:
        value, shapes, lsi, loc, aw = config2(B)
        go = torch.randn(B, 150, 512, device="cuda")
        bf16 = args.dtype == "bf16"
        if bf16:  # the training step's layout: bf16 value map / output rows, fp32 sampling geometry
            value, go = value.bfloat16(), go.bfloat16()
        tf = time_gpu(lambda: ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64), args.iters)
        tb = time_gpu(lambda: ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64), args.iters)
        # backward wrapper also zero-fills grad_value: time the memset separately
        tz = time_gpu(lambda: torch.zeros_like(value), args.iters)
        fb, bb = msda_bytes(B, bf16)
        res.append({"B": B, "dtype": args.dtype, "fwd_algorithmic_MB": fb / 1e6, "bwd_algorithmic_MB": bb / 1e6, "fwd_us": tf * 1e6, "fwd_GBps": fb / tf / 1e9, "fwd_frac_8TBps": fb / tf / 8e12,
                    "bwd_us(incl zero-fill)": tb * 1e6, "zero_fill_us": tz * 1e6, "bwd_GBps": bb / tb / 1e9,
                    "bwd_frac_8TBps": bb / tb / 8e12})
        print(json.dumps(res[-1]))
    return res


def bench_winattn(args):
    """Swin stages of the 640x640 benchmark at batch 32: (map side, heads)."""
    from grit_amd.ops.window_attention import window_attention
    res = []
    for (side, nH) in [(160, 4), (80, 8), (40, 16), (20, 32)]:
        B = args.batch[-1]
        C = 32 * nH
        qkv = torch.randn(B, side * side, 3 * C, device="cuda").bfloat16().requires_grad_(True)
        bias = (torch.randn(nH, 144, 144, device="cuda") * 0.5).requires_grad_(True)
        pad = torch.randn(3 * C, device="cuda").bfloat16().requires_grad_(True)
        for shift in (0, 6):
            f = lambda: window_attention(qkv, bias, pad, side, side, nH, 12, shift, 32**-0.5)
            tf = time_gpu(f, args.iters)
            out = f()
            g = torch.randn_like(out)
            def fb():
                qkv.grad = bias.grad = pad.grad = None
                window_attention(qkv, bias, pad, side, side, nH, 12, shift, 32**-0.5).backward(g)
            tfb = time_gpu(fb, max(10, args.iters // 4))
            nwin = B * (-(-side // 12))**2
            flops_f = nwin * nH * 4 * 144 * 144 * 32
            io = B * side * side * 4 * C * 2
            res.append({"side": side, "heads": nH, "shift": shift, "fwd_us": tf * 1e6, "fwd+bwd_us": tfb * 1e6,
                        "fwd_TFLOPs": flops_f / tf / 1e12, "fwd_io_GBps": io / tf / 1e9,
                        "ns_per_window_head_fwd": tf * 1e9 / (nwin * nH), "ns_per_window_head_bwd": (tfb - tf) * 1e9 / (nwin * nH)})
            print(json.dumps(res[-1]))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["msda", "winattn", "attn"])
    ap.add_argument("--batch", type=int, nargs="+", default=[8, 32])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="msda: dtype of the value map")
    a = ap.parse_args()
    {"msda": bench_msda, "winattn": bench_winattn}[a.which](a)
