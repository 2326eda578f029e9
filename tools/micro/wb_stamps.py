"""Phase stamps of winattn_bwd_dma: cycles per workgroup, wave and phase.  Needs the diagnostic build of
tools/micro/wb_stamps_patch.py (apply -> run this on the GPU box -> restore)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from grit_amd import lib
from grit_amd.ops.window_attention import window_attention
L = lib.LIB if hasattr(lib, "LIB") else lib.load()
f = L.grit_debug_winattn_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_int]; f.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 256)()
names = ["loop-top", "vmcnt0", "barrier1", "delta+frags", "barrier2", "prefetch", "phase1", "dkdv store+pad", "barrier3", "phase2+stores"]
print(' '.join(names))
for (side, nH) in [(160, 4), (40, 16)]:
    B, C = 32, 32 * nH
    qkv = torch.randn(B, side * side, 3 * C, device="cuda").bfloat16().requires_grad_(True)
    bias = (torch.randn(nH, 144, 144, device="cuda") * 0.5).requires_grad_(True)
    pad = torch.randn(3 * C, device="cuda").bfloat16().requires_grad_(True)
    out = window_attention(qkv, bias, pad, side, side, nH, 12, 6, 32**-0.5)
    g = torch.randn_like(out)
    out.backward(g); torch.cuda.synchronize()
    f(buf, 1)
    out = window_attention(qkv, bias, pad, side, side, nH, 12, 6, 32**-0.5)
    out.backward(g); torch.cuda.synchronize()
    f(buf, 1)
    for base, nm in [(16 * w, "wave%d" % w) for w in range(9)]:
        n = buf[base + 10]
        tot = sum(buf[base + k] for k in range(10))
        print(side, nm, tot // max(n, 1), " ".join("%7d" % (buf[base + k] // max(n, 1)) for k in range(10)))
