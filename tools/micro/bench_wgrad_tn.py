"""grit_wgrad_tn (256 x 256 tiles, transposing LDS reads) against the library path it would replace (batched GEMM over 16 row slices
with fp32 partials) on the Swin weight-gradient shapes; both followed by the same grouped slab sum.  us per call, HIP events."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd import lib as _lib  # noqa: E402
from grit_amd.ops.linear import slab_sum, split_k  # noqa: E402


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    if os.environ.get("GRIT_TUNED_GEMMS", "1") == "1":
        import bench
        bench._enable_tuned_gemms()
    lib = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    only = os.environ.get("SHAPES")
    for (M, N, K) in [tuple(int(v) for v in sh.split("x")) for sh in only.split(",")] if only else [(51200, 2048, 512), (51200, 512, 2048), (51200, 1536, 512), (51200, 512, 512), (12800, 4096, 1024),
                      (12800, 1024, 4096), (204800, 1024, 256), (204800, 256, 1024), (204800, 768, 256), (204800, 256, 256)]:
        g = torch.Generator(device="cuda").manual_seed(0)
        dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        S0 = split_k(M)
        lib_gemm = lambda: torch.bmm(dy.view(S0, M // S0, N).transpose(1, 2), x.view(S0, M // S0, K), out_dtype=torch.float32)
        part0 = lib_gemm()
        ref = slab_sum(part0.unsqueeze(0), torch.float32)[0]
        t_lib, t_lib_sum = timed(lib_gemm), timed(lambda: slab_sum(part0.unsqueeze(0), torch.bfloat16))
        S = lib.grit_wgrad_tn_splits(M, N, K)
        if S <= 0:
            print(f"M{M} N{N} K{K}: library S{S0} {t_lib:.0f} + sum {t_lib_sum:.0f} us | own: shape not supported", flush=True)
            continue
        part = torch.empty(S, N, K, dtype=torch.float32, device="cuda")
        bpart = torch.empty(S, N, dtype=torch.float32, device="cuda")
        own = lambda: lib.grit_wgrad_tn(p(dy), N, p(x), K, M, N, K, S, p(part), None, _lib.current_stream_ptr())
        own_b = lambda: lib.grit_wgrad_tn(p(dy), N, p(x), K, M, N, K, S, p(part), p(bpart), _lib.current_stream_ptr())
        assert own() == 0
        got = slab_sum(part.unsqueeze(0), torch.float32)[0]
        err = float((got - ref).abs().max() / ref.abs().max())
        t_own, t_own_sum = timed(own), timed(lambda: slab_sum(part.unsqueeze(0), torch.bfloat16))
        t_own_b = timed(own_b)
        if int(os.environ.get("GRIT_WGRAD_TN_DBG", "0")) & 16:
            own()
            torch.cuda.synchronize()
            print(f"    counter cycles per 64-row K step (workgroup 0, wave 0): {float(part[0, 0, 0]):.0f}; {M // 64 // S} steps per workgroup"
                  f" -> loop {float(part[0, 0, 0]) * (M // 64 // S):.0f} cycles of a {t_own:.0f} us launch", flush=True)
        fl = 2.0 * M * N * K
        print(f"M{M} N{N} K{K}: library S{S0} {t_lib:.0f} us ({fl / t_lib / 1e9:.2f} PF/s) + sum {t_lib_sum:.0f} | own S{S} {t_own:.0f} us "
              f"({fl / t_own / 1e9:.2f} PF/s) + sum {t_own_sum:.0f} | with bias by-product {t_own_b:.0f} us | rel err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
