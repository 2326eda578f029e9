"""Short-map Linears of the two decoders (640 .. 12 800 rows): the small-tile instantiations of grit_gemm_bf16_nt (variants 10-13) against
the library (tuned table) -- forward form y = x W^T + b.  HIP events around loops of 50 calls, arms interleaved, best of 2 rounds."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd.ops import gemm as G


def t(fn, it=50):
    for _ in range(5):
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
    shapes = [(4800, 512, 512), (4800, 1024, 512), (4800, 512, 1024), (4800, 256, 512), (4800, 128, 512), (4800, 512, 256), (4800, 512, 128),
              (640, 512, 512), (640, 2048, 512), (640, 512, 2048), (640, 1024, 512), (640, 512, 1024),
              (12800, 512, 1024), (12800, 512, 512), (12800, 1024, 512), (3200, 512, 512), (8500 * 32 // 64, 512, 512)]
    # graph replay timing too: the step replays its kernels back to back, no host launch cost
    for M, N, K in shapes:
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
        b = torch.randn(N, device='cuda').bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        ref = F.linear(x, w, b).float()
        arms = [("lib", lambda: F.linear(x, w, b))]
        for v in (10, 11, 12, 13, 1):
            try:
                G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=v)
                err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
                if err > 2e-2:
                    print("variant", v, "WRONG", err)
                arms.append(("v%d" % v, (lambda v=v: G.gemm_nt(x, w, G.BIAS, bias=b, out=out, variant=v))))
            except Exception as e:
                pass
        res = {}
        for rnd in range(2):
            for name, fn in arms:
                # inside a graph: 20 back-to-back launches
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    fn()
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=s):
                        for _ in range(20):
                            fn()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(t(g.replay, it=10) / 20)
        best = {k: min(v) for k, v in res.items()}
        print("M%-6d N%-5d K%-5d  " % (M, N, K) + "  ".join("%s %5.1f" % (k, v) for k, v in best.items()), flush=True)


if __name__ == "__main__":
    main()
