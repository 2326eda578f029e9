"""Window attention forward / backward with and without the drop-path factors (stage-2 geometry of the benchmark: 32 images, 40 x 40
tokens, 16 heads), HIP events around loops of 20 launches."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd import lib as _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)  # noqa: E731


def bench(B, H, W, nH, dropped):
    C = 32 * nH
    qkv = torch.randn(B, H * W, 3 * C, device=DEV).bfloat16()
    bias = torch.randn(nH, 144, 144, device=DEV)
    pad = torch.randn(3 * C, device=DEV).bfloat16()
    out = torch.empty(B, H * W, C, device=DEV, dtype=torch.bfloat16)
    nW = B * (-(-H // 12)) * (-(-W // 12))
    lse = torch.empty(nW, nH, 144, device=DEV)
    dout = torch.randn(B, H * W, C, device=DEV).bfloat16()
    scale = torch.full((B,), 1.2, device=DEV)
    for b in dropped:
        scale[b] = 0
        dout[b] = 0
    dqkv = torch.empty_like(qkv)
    acc = torch.zeros(bias.numel() + 3 * C, device=DEV)
    st = _lib.current_stream_ptr()

    def fwd(rs):
        return lib.grit_winattn_fwd_bf16_rows(p(qkv), p(bias), p(pad), None, 0, B, H, W, C, nH, 12, 6, 32 ** -0.5, p(out), p(lse), p(rs) if rs is not None else None, st)

    def bwd(rs):
        return lib.grit_winattn_bwd_bf16_rows(p(qkv), p(bias), p(pad), None, 0, p(out), p(dout), p(lse), B, H, W, C, nH, 12, 6, 32 ** -0.5,
                                              p(dqkv), p(acc[:bias.numel()]), p(acc[bias.numel():]), p(rs) if rs is not None else None, st)

    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for rs in (None, scale):
            for _ in range(3):
                assert fn(rs) == 0
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn(rs)
            b.record()
            torch.cuda.synchronize()
            print(f"B {B} {H}x{W} heads {nH} dropped {len(dropped)}: {name} factors {'on ' if rs is not None else 'off'} {a.elapsed_time(b) / 20 * 1e3:8.1f} us", flush=True)


bench(32, 40, 40, 16, (3, 7, 12, 20, 29))
bench(32, 40, 40, 16, ())
bench(32, 80, 80, 8, (1, 5, 9, 30))
bench(32, 20, 20, 32, (2, 4, 8))
