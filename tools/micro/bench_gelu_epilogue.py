"""Is the library's GELU epilogue free?  fc1 shapes of the four Swin stages at batch 32 x 640^2, bf16, one loop per variant."""
import sys, os
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd.ops import gemm as G


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (M, C) in [(819200, 128), (204800, 256), (51200, 512), (12800, 1024)]:
    x = torch.randn(M, C, device='cuda').bfloat16()
    w = (torch.randn(4 * C, C, device='cuda') * 0.05).bfloat16()
    b = torch.randn(4 * C, device='cuda').bfloat16()
    t_lin = timeit(lambda: F.linear(x, w, b))
    t_lin_gelu = timeit(lambda: F.gelu(F.linear(x, w, b)))
    try:
        t_act = timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True))
        y = torch._addmm_activation(b, x, w.t(), use_gelu=True)
        ref = F.gelu(F.linear(x, w, b).float())
        ref_t = F.gelu(F.linear(x, w, b).float(), approximate='tanh')
        err = (y.float() - ref).abs().max().item(), (y.float() - ref_t).abs().max().item()
    except Exception as e:
        t_act, err = float('nan'), str(e)[:80]
    t_own = timeit(lambda: G.linear_bias_gelu(x, w, b))
    t_own_noaux = timeit(lambda: G.gemm_nt(x, w, G.BIAS_GELU, bias=b))
    print("M %7d C %4d: linear %6.1f us | linear+gelu kernels %6.1f | _addmm_activation(gelu) %6.1f (max err vs erf / tanh: %s) | own BIAS_GELU+aux %6.1f | own no aux %6.1f"
          % (M, C, t_lin, t_lin_gelu, t_act, err, t_own, t_own_noaux))
