"""grit_wgrad_small (+ grouped reduction) against the library's transposed GEMM + column sum on the short-map Linear shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grit_amd.tuning import load_tuned_gemms
from grit_amd.ops import linear as L
load_tuned_gemms()


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for M, N, K in [(4800, 512, 512), (4800, 1024, 512), (4800, 512, 1024), (4800, 256, 512), (4800, 128, 512), (3200, 512, 512),
                (3200, 2048, 512), (3200, 512, 2048), (3200, 512, 1024), (640, 512, 512), (640, 2048, 512), (640, 512, 2048)]:
    dy = torch.randn(M, N, device='cuda').bfloat16()
    x = torch.randn(M, K, device='cuda').bfloat16()
    lib_t = timeit(lambda: torch.mm(dy.t(), x))
    lib_b = timeit(lambda: L.column_sum(dy, torch.bfloat16))
    own = timeit(lambda: L.small_weight_bias_grad(dy, x, True, torch.bfloat16))
    import ctypes
    from grit_amd import lib as _lib
    lib = _lib.load()
    S = lib.grit_wgrad_small_splits(M, N, K)
    wp = torch.empty(S, N, K, device="cuda", dtype=torch.bfloat16 if S == 1 else torch.float32); bp = torch.empty(S, N, device="cuda", dtype=wp.dtype)
    raw = timeit(lambda: lib.grit_wgrad_small(ctypes.c_void_p(dy.data_ptr()), N, ctypes.c_void_p(x.data_ptr()), K, M, N, K, S,
                                               ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(bp.data_ptr()), _lib.current_stream_ptr()))
    print(f"M {M:5d} N {N:5d} K {K:5d} splits {S:3d}: library mm {lib_t:6.1f} us + colsum/slab {lib_b:6.1f} us | own kernel {raw:6.1f} us, with reduction {own:6.1f} us")
