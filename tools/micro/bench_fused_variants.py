"""The fused Mlp GEMMs (fc1 + bias + GELU with both outputs; fc2 input gradient x GELU' + column sums) across the tile variants of
grit_gemm_bf16_nt (1: 256x128x32 three-slot ring, two workgroups per CU; 2: 256x128x64; 3: 256x128x32 four slots; 4: 256x256x64
eight waves (the default); 5: persistent ping-pong; 7: four waves) on the Swin shapes; HIP events around loops of 20 calls."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from grit_amd.ops import gemm as G
from bench_w4_vs_lib import t


def main():
    torch.manual_seed(0)
    for M, C in ((819200, 128), (204800, 256), (51200, 512), (12800, 1024)):
        x = torch.randn(M, C, device='cuda').bfloat16()
        w1 = (torch.randn(4 * C, C, device='cuda') * C ** -0.5).bfloat16()
        b1 = torch.randn(4 * C, device='cuda').bfloat16()
        pre = torch.empty(M, 4 * C, device='cuda', dtype=torch.bfloat16)
        act = torch.empty_like(pre)
        dy = torch.randn(M, C, device='cuda').bfloat16()
        w2t = (torch.randn(4 * C, C, device='cuda') * (4 * C) ** -0.5).bfloat16()  # fc2.weight^T: [4C, C]
        dpre = torch.empty_like(pre)
        part = torch.empty((-(-M // 128), 4 * C), dtype=torch.float32, device='cuda')
        G.gemm_nt(x, w1, G.BIAS_GELU, bias=b1, aux=pre, out=act, variant=4)
        line = "M%-7d C%-5d" % (M, C)
        for v in (1, 2, 3, 4, 5, 7):
            try:
                tg = t(lambda: G.gemm_nt(x, w1, G.BIAS_GELU, bias=b1, aux=pre, out=act, variant=v))
                td = t(lambda: G.gemm_nt(dy, w2t, G.DGELU, aux=pre, colsum=part, out=dpre, variant=v))
                line += " | v%d gelu %6.1f dgelu %6.1f" % (v, tg, td)
            except Exception as e:
                line += " | v%d: %s" % (v, str(e)[:30])
        print(line, flush=True)


if __name__ == "__main__":
    main()
