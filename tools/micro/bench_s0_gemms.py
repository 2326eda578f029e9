"""Stage-0 long-map GEMMs (819 200 tokens, C = 128): the library's NT form against the HBM bound (operands once + output once at
8 TB/s) and against the eight-wave own kernel where its tiling applies; HIP events around loops of 20 calls."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd.ops import gemm as G
from bench_w4_vs_lib import t


def main():
    import bench
    print("tuned table:", bench._enable_tuned_gemms())
    torch.manual_seed(0)
    M = int(os.environ.get("ROWS", 819200))
    for name, N, K in (("qkv fwd", 384, 128), ("proj fwd / dgrad", 128, 128), ("fc2 fwd / fc1 dgrad", 128, 512), ("qkv dgrad", 128, 384),
                       ("fc1 fwd (bias only)", 512, 128)):
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
        b = torch.randn(N, device='cuda').bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        floor = (M * K + N * K + M * N) * 2 / 8e12 * 1e6
        t_lib = t(lambda: F.linear(x, w, b))
        t_nb = t(lambda: F.linear(x, w))
        line = "%-22s M%-7d N%-4d K%-4d  HBM floor %6.1f us | lib NT+bias %6.1f (%.2f)  no bias %6.1f" % (name, M, N, K, floor, t_lib, floor / t_lib, t_nb)
        for v in (0, 7):
            try:
                G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=v)
                tv = t(lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=v))
                line += " | own v%d %6.1f (%.2f)" % (v, tv, floor / tv)
            except Exception as e:
                line += " | own v%d: %s" % (v, str(e)[:40])
        print(line, flush=True)


if __name__ == "__main__":
    main()
