"""grit_gemm_bf16_nt against the library path it replaces (F.linear [+ gelu / gelu_backward + column sum]): correctness on
the Swin shapes and time per call (HIP events around a loop).  `python tools/bench_gemm.py [variants...]`"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grit_amd.ops import gemm as G
from grit_amd.ops.linear import column_sum, slab_sum


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


def relerr(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20)).item()


def main():
    variants = [int(v) for v in sys.argv[1:]] or [1]
    if os.environ.get("GRIT_TUNED_GEMMS", "1") == "1":
        import bench
        print("tuned table:", bench._enable_tuned_gemms())
    torch.manual_seed(0)
    shapes = [(51200, 2048, 512), (51200, 1536, 512), (51200, 512, 512), (51200, 512, 2048), (204800, 1024, 256),
              (12800, 4096, 1024), (819200, 512, 128), (51200 - 100, 2048, 512)]
    for (M, N, K) in shapes:
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
        b = torch.randn(N, device='cuda').bfloat16()
        flop = 2.0 * M * N * K
        ref_pre = F.linear(x, w, b)
        ref_act = F.gelu(ref_pre)
        t_lin = t(lambda: F.linear(x, w, b))
        t_gelu = t(lambda: F.gelu(ref_pre))
        line = [f"M{M} N{N} K{K}: lib linear {t_lin:.0f}us ({flop / t_lin / 1e9:.0f} TF) gelu {t_gelu:.0f}us"]
        for v in variants:
            try:
                out = G.gemm_nt(x, w, G.BIAS, bias=b, variant=v)
                e1 = relerr(out, ref_pre)
                pre = torch.empty_like(out)
                act = G.gemm_nt(x, w, G.BIAS_GELU, bias=b, aux=pre, variant=v)
                e2, e3 = relerr(pre, ref_pre), relerr(act, F.gelu(ref_pre.float()))
                t1 = t(lambda: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=v))
                t2 = t(lambda: G.gemm_nt(x, w, G.BIAS_GELU, bias=b, aux=pre, out=act, variant=v))
                line.append(f"| v{v} bias {t1:.0f}us ({flop / t1 / 1e9:.0f} TF, err {e1:.1e}) bias+gelu {t2:.0f}us (err {e2:.1e} {e3:.1e})")
            except Exception as e:
                line.append(f"| v{v} {str(e)[:60]}")
        print(" ".join(line), flush=True)
    # backward pair: d_pre = (dy @ W2) * gelu'(pre), db1 = colsum(d_pre)   (fc2 input gradient of a Swin Mlp)
    for (M, C) in [(51200, 512), (204800, 256), (12800, 1024), (51200 - 100, 512)]:
        H = 4 * C
        dy = torch.randn(M, C, device='cuda').bfloat16()
        w2 = (torch.randn(C, H, device='cuda') * H ** -0.5).bfloat16()  # fc2.weight [C, 4C]
        pre = torch.randn(M, H, device='cuda').bfloat16()
        w2t = w2.t().contiguous()

        def lib():
            dact = torch.mm(dy, w2)
            dpre = torch.ops.aten.gelu_backward(dact, pre)
            return dpre, column_sum(dpre, torch.bfloat16)
        ref_dpre = torch.ops.aten.gelu_backward(torch.mm(dy.float(), w2.float()), pre.float())
        ref_db = ref_dpre.sum(0)
        t_lib = t(lib)
        line = [f"dgelu M{M} C{C}: lib mm+gelu_bwd+colsum {t_lib:.0f}us"]
        for v in variants:
            try:
                dpre, part = G.input_grad_dgelu(dy, w2t, pre)
                db = slab_sum(part.unsqueeze(0), torch.float32)[0]
                e1, e2 = relerr(dpre, ref_dpre), relerr(db, ref_db)
                G.VARIANT = v
                t1 = t(lambda: G.input_grad_dgelu(dy, w2t, pre))
                line.append(f"| v{v} fused {t1:.0f}us (err dpre {e1:.1e} db {e2:.1e})")
            except Exception as e:
                line.append(f"| v{v} {str(e)[:60]}")
        print(" ".join(line), flush=True)


if __name__ == "__main__":
    main()
