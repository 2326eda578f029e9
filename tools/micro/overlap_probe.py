"""Does matrix work overlap with HBM streaming on this chip, or do the two add up (power / clock give-back)?  A compute-bound bf16 GEMM
(4096 x 4096 x 65536: 2.2 TFLOP, 0.6 GB of operands) and an HBM-bound copy (2 x 2 GiB) alone and side by side on two streams; the GEMM
on half / all CUs is not controllable from here, so the copy is sized to run as long as the GEMM."""
import torch
import time

def ev():
    return torch.cuda.Event(enable_timing=True)

torch.manual_seed(0)
a = torch.randn(4096, 65536, device='cuda').bfloat16()
b = torch.randn(4096, 65536, device='cuda').bfloat16()
src = torch.empty(1 << 30, dtype=torch.int16, device='cuda').random_()
dst = torch.empty_like(src)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def gemm(n):
    for _ in range(n):
        torch.mm(a, b.t())

def copy(n):
    for _ in range(n):
        dst.copy_(src)

def timed(fn_a, fn_b=None):
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        fn_a()
    if fn_b is not None:
        with torch.cuda.stream(s2):
            fn_b()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)

for _ in range(2):
    gemm(2); copy(2)
tg = timed(lambda: gemm(10))
tc = timed(lambda: copy(10))
nc = max(1, round(10 * tg / tc))
tc = timed(lambda: copy(nc))
both = timed(lambda: gemm(10), lambda: copy(nc))
print("GEMM x10 alone %.2f ms (%.0f TFLOP/s) | copy x%d alone %.2f ms (%.2f TB/s) | both on two streams %.2f ms | sum %.2f  max %.2f" %
      (tg, 10 * 2 * 4096 * 4096 * 65536 / tg / 1e9, nc, tc, nc * 2 * src.numel() * 2 / tc / 1e9, both, tg + tc, max(tg, tc)))
