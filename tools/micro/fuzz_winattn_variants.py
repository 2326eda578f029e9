"""Window attention forward + backward on 36 random geometries; prints a digest per geometry.  Run once per kernel variant
(GRIT_WINATTN_FWD_DMA / GRIT_WINATTN_BWD_DMA) and compare the lines: tests/test_winattn_gpu.py does."""
import sys, torch, hashlib, random
sys.path.insert(0, ".")
from grit_amd.ops.window_attention import window_attention
random.seed(1)
geos = []
for _ in range(36):
    nH = random.choice([1, 2, 3, 4, 8, 16])
    geos.append((random.randint(1, 3), random.randint(5, 50), random.randint(5, 50), nH, random.choice([0, 6])))
out = []
for (B, H, W, nH, shift) in geos:
    g = torch.Generator().manual_seed(H * 100 + W)
    C = 32 * nH
    qkv = torch.randn(B, H * W, 3 * C, generator=g).bfloat16().cuda().requires_grad_(True)
    bias = (torch.randn(nH, 144, 144, generator=g) * 0.5).bfloat16().float().cuda().requires_grad_(True)  # bf16-representable: the DMA backward keeps the slab in bf16
    pad = (torch.randn(3 * C, generator=g) * 0.3).bfloat16().cuda().requires_grad_(True)
    cot = torch.randn(B, H * W, C, generator=g).bfloat16().cuda()
    o = window_attention(qkv, bias, pad, H, W, nH, 12, shift, 32**-0.5)
    o.backward(cot)
    torch.cuda.synchronize()
    h = hashlib.sha1(o.detach().float().cpu().numpy().tobytes()).hexdigest()[:12]
    hq = hashlib.sha1(qkv.grad.float().cpu().numpy().tobytes()).hexdigest()[:12]
    out.append("%s fwd %s dqkv %s dbias %.6e dpad %.6e" % ((B, H, W, nH, shift), h, hq, bias.grad.double().abs().sum().item(), pad.grad.double().abs().sum().item()))
print("\n".join(out))
