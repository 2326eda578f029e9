"""Where the time of fc1 + bias + GELU (M 51 200, N 2 048) goes: the same launch at K = 64 .. 512 (MFMA work scales with K, the 420 MB of
output and the GELU arithmetic do not) with two outputs, one output + GELU, one output without GELU (bias only)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from grit_amd.ops import gemm as G
from bench_w4_vs_lib import t

M, N = 51200, 2048
torch.manual_seed(0)
for K in (64, 128, 256, 512, 1024):
    x = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
    b = torch.randn(N, device='cuda').bfloat16()
    pre = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    act = torch.empty_like(pre)
    line = "K %-5d MFMA-only floor %5.1f us (1.3 PF) |" % (K, 2.0 * M * N * K / 1.3e15 * 1e6)
    for v in (4, 7):
        two = t(lambda: G.gemm_nt(x, w, G.BIAS_GELU, bias=b, aux=pre, out=act, variant=v))
        one = t(lambda: G.gemm_nt(x, w, G.BIAS_GELU, bias=b, out=act, variant=v))
        bias = t(lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=act, variant=v))
        line += "  v%d: gelu 2 maps %6.1f  gelu 1 map %6.1f  bias 1 map %6.1f |" % (v, two, one, bias)
    print(line, flush=True)
