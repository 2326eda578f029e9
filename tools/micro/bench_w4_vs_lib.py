"""The four-wave persistent GEMM (grit_gemm_bf16_nt variant 7) against the library on every long-map Linear shape of the step
(forward: y = x W^T + b; input gradient: dx = dy W as NT on the transposed weight), HIP events around loops of 20 calls."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd.ops import gemm as G


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    import bench
    print("tuned table:", bench._enable_tuned_gemms())
    torch.manual_seed(0)
    rows = {0: 819200, 1: 204800, 2: 51200, 3: 12800}
    shapes = []
    for st, C in ((0, 128), (1, 256), (2, 512), (3, 1024)):
        M = rows[st]
        shapes += [("qkv fwd s%d" % st, M, 3 * C, C), ("proj fwd/dgrad s%d" % st, M, C, C), ("fc2 fwd / fc1 dgrad s%d" % st, M, C, 4 * C),
                   ("qkv dgrad s%d" % st, M, C, 3 * C), ("fc1 fwd (bias only) s%d" % st, M, 4 * C, C)]
    shapes += [("value_proj stacked", 272000, 3072, 512), ("value_proj dgrad", 272000, 512, 3072)]
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
        b = torch.randn(N, device='cuda').bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        ref = F.linear(x, w, b)
        t_lib = t(lambda: F.linear(x, w, b))
        wt = w.t().contiguous()
        t_mm = t(lambda: torch.mm(x, wt))  # the library's NN form (what the input gradients run)
        try:
            G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=7)
            err = ((out.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
            t_own = t(lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=7))
            t_v0 = t(lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=0))
            t_v9 = t(lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=9))
            t_lib2 = t(lambda: F.linear(x, w, b))
            print("%-26s M%-7d N%-5d K%-5d  lib NT %6.1f / %6.1f  lib NN %6.1f  own v7 %6.1f  v9 (%d rows) %6.1f us  (%.2f x NT)  err %.1e  | eight-wave v0 %6.1f" %
                  (name, M, N, K, t_lib, t_lib2, t_mm, t_own, G.w4_tile_rows(M, N), t_v9, min(t_lib, t_lib2) / t_v9, err, t_v0), flush=True)
        except Exception as e:
            try:
                t_v0 = t(lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=0))
            except Exception:
                t_v0 = float("nan")
            print("%-26s M%-7d N%-5d K%-5d  lib NT %6.1f  lib NN %6.1f  own: %s  | eight-wave v0 %6.1f" % (name, M, N, K, t_lib, t_mm, str(e)[:40], t_v0), flush=True)


if __name__ == "__main__":
    main()
