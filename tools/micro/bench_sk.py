"""Stream-K form of the four-wave persistent GEMM (grit_gemm_bf16_nt_sk) against the plain variant 7 and the library on the long-map
shapes whose tiles do not fill the last round; HIP events around loops of 20 calls, the three arms interleaved per shape."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from grit_amd.ops import gemm as G
from bench_w4_vs_lib import t


def main():
    import bench
    print("tuned table:", bench._enable_tuned_gemms())
    torch.manual_seed(0)
    shapes = [("fc2 fwd / fc1 dgrad s2", 51200, 512, 2048), ("qkv dgrad s2", 51200, 512, 1536), ("proj s2", 51200, 512, 512),
              ("qkv fwd s2", 51200, 1536, 512), ("fc1 bias-only s2", 51200, 2048, 512),
              ("fc2 fwd / fc1 dgrad s1", 204800, 256, 1024), ("qkv dgrad s1", 204800, 256, 768), ("proj s1", 204800, 256, 256),
              ("fc2 fwd / fc1 dgrad s3", 12800, 1024, 4096), ("qkv dgrad s3", 12800, 1024, 3072), ("proj s3", 12800, 1024, 1024),
              ("qkv fwd s3", 12800, 3072, 1024), ("fc1 bias-only s3", 12800, 4096, 1024),
              ("value_proj dgrad", 272000, 512, 3072), ("merge s1->s2", 51200, 512, 1024), ("merge s2->s3", 12800, 1024, 2048)]
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
        b = torch.randn(N, device='cuda').bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        res = {}
        for rnd in range(2):
            for arm, fn in (("lib", lambda: F.linear(x, w, b)),
                            ("w4", lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=7)),
                            ("sk", lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=G.SK))):
                try:
                    res.setdefault(arm, []).append(t(fn))
                except Exception as e:
                    res.setdefault(arm, []).append(float("nan"))
        best = {k: min(v) for k, v in res.items()}
        print("%-26s M%-7d N%-5d K%-5d  lib %6.1f  w4 %6.1f  sk %6.1f us   sk/lib %.2f  sk/w4 %.2f" %
              (name, M, N, K, best["lib"], best["w4"], best["sk"], best["sk"] / best["lib"], best["sk"] / best["w4"]), flush=True)


if __name__ == "__main__":
    main()
