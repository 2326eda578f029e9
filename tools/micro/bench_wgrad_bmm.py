"""Headroom of the split-M weight-gradient bmm: fp32-out (what runs; not covered by TunableOp) vs bf16-out default vs bf16-out tuned."""
import os, sys, torch
import torch.cuda.tunable as tunable
torch.manual_seed(0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shapes = [(51200, 2048, 512), (51200, 512, 2048), (51200, 1536, 512), (51200, 512, 512), (204800, 1024, 256), (12800, 4096, 1024)]
S_of = lambda M: 16 if M >= 51200 else 4
res = {}
for phase in ("default", "tuned"):
    if phase == "tuned":
        tunable.enable(True); tunable.tuning_enable(True); tunable.set_max_tuning_duration(60); tunable.set_max_tuning_iterations(30)
        tunable.set_filename("/tmp/bmm_tune.csv")
    for (M, N, K) in shapes:
        S = min(64, M // 3200) if M >= 25600 else 4
        dy = torch.randn(M, N, device='cuda').bfloat16(); x = torch.randn(M, K, device='cuda').bfloat16()
        a = dy.view(S, M // S, N).transpose(1, 2); b = x.view(S, M // S, K)
        t32 = timeit(lambda: torch.bmm(a, b, out_dtype=torch.float32))
        t16 = timeit(lambda: torch.bmm(a, b))
        tmm = timeit(lambda: torch.mm(dy.t(), x))
        print(f"{phase:8s} M {M} N {N} K {K} S {S}: bmm fp32-out {t32:6.1f} us | bmm bf16-out {t16:6.1f} us | plain mm (library split) {tmm:6.1f} us")
if os.path.exists("/tmp/bmm_tune.csv"):
    print(open("/tmp/bmm_tune.csv").read()[-1500:])
