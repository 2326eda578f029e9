"""Does the four-wave GEMM run faster when 16 % of the rows of its A operand are zeros (rows of samples drop path removed)?  The matrix
pipe is power-bound (profiles/r05/wgrad_rows_microbench.txt): zeros are cheaper to multiply."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd.ops import gemm as G  # noqa: E402

DEV = "cuda"
for (M, N, K, name) in ((51200, 1536, 512, "stage-2 qkv fwd"), (51200, 512, 512, "stage-2 proj fwd"), (51200, 512, 2048, "stage-2 fc2 fwd (library shape)")):
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=DEV).bfloat16()
    for frac in (0.0, 0.16, 0.34):
        x = torch.randn(M, K, device=DEV).bfloat16()
        per = 1600
        for s in range(int(round(frac * 32))):
            x[(3 * s % 32) * per:(3 * s % 32 + 1) * per] = 0
        for label, fn in (("own w4", lambda: G.gemm_nt(x, w, G.BIAS, bias=b, variant=7)), ("library", lambda: torch.nn.functional.linear(x, w, b))):
            for _ in range(3):
                fn()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            print(f"{name}: {label:8s} zero rows {frac:4.2f}: {a.elapsed_time(e) / 20 * 1e3:7.1f} us", flush=True)
