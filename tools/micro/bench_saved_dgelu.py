"""fc1 + GELU saving the pre-activation (BIAS_GELU) or the derivative (BIAS_GELU_DACT), and the fc2 input gradient with the GELU' epilogue
(DGELU) or the plain product (MUL_AUX), eight-wave kernel, Swin shapes; HIP events around loops of 20 calls."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from grit_amd.ops import gemm as G
from bench_w4_vs_lib import t


def main():
    torch.manual_seed(0)
    for M, C in ((204800, 256), (51200, 512), (12800, 1024)):
        x = torch.randn(M, C, device='cuda').bfloat16()
        w1 = (torch.randn(4 * C, C, device='cuda') * C ** -0.5).bfloat16()
        b1 = torch.randn(4 * C, device='cuda').bfloat16()
        pre = torch.empty(M, 4 * C, device='cuda', dtype=torch.bfloat16)
        act = torch.empty_like(pre)
        dy = torch.randn(M, C, device='cuda').bfloat16()
        w2t = (torch.randn(4 * C, C, device='cuda') * (4 * C) ** -0.5).bfloat16()
        dpre = torch.empty_like(pre)
        part = torch.empty((-(-M // 128), 4 * C), dtype=torch.float32, device='cuda')
        G.gemm_nt(x, w1, G.BIAS_GELU, bias=b1, aux=pre, out=act)
        r = {}
        r["gelu+pre"] = t(lambda: G.gemm_nt(x, w1, G.BIAS_GELU, bias=b1, aux=pre, out=act))
        r["gelu+dact"] = t(lambda: G.gemm_nt(x, w1, G.BIAS_GELU_DACT, bias=b1, aux=pre, out=act))
        r["bias only"] = t(lambda: G.gemm_nt(x, w1, G.BIAS, bias=b1, out=act))
        r["dgelu"] = t(lambda: G.gemm_nt(dy, w2t, G.DGELU, aux=pre, colsum=part, out=dpre))
        r["mul_aux"] = t(lambda: G.gemm_nt(dy, w2t, G.MUL_AUX, aux=pre, colsum=part, out=dpre))
        r["none"] = t(lambda: G.gemm_nt(dy, w2t, G.NONE, out=dpre))
        print("M%-7d C%-5d " % (M, C) + "  ".join("%s %6.1f" % kv for kv in r.items()), flush=True)


if __name__ == "__main__":
    main()
