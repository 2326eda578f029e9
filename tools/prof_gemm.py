"""Run grit_gemm_bf16_nt a few times for rocprofv3 (kernel trace / PMC passes): python3 tools/prof_gemm.py M N K epilogue variant iters"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grit_amd.ops import gemm as G

M, N, K, epi, variant, iters = (int(v) for v in (sys.argv[1:] + ["51200", "2048", "512", "1", "1", "10"])[:6])
torch.manual_seed(0)
x = torch.randn(M, K, device='cuda').bfloat16()
w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
b = torch.randn(N, device='cuda').bfloat16()
out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
aux = torch.randn(M, N, device='cuda').bfloat16()
part = torch.empty(-(-M // 128), N, device='cuda', dtype=torch.float32)
for _ in range(iters):
    G.gemm_nt(x, w, epi, bias=b, aux=aux, colsum=part, out=out, variant=variant)
torch.cuda.synchronize()
