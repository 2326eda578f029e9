"""Microbenchmark behind grit_amd/ops/linear.py: dW = dY^T X as one torch.mm vs split-M batched GEMMs (fp32 partials)."""
import torch, time
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
for (M,K,N) in [(51200,512,2048),(51200,2048,512),(51200,512,1536),(51200,512,512),(204800,256,1024),(204800,1024,256),(204800,256,768),(204800,256,256),(12800,1024,4096),(272000,512,512)]:
    x=torch.randn(M,K,device='cuda').bfloat16(); dy=torch.randn(M,N,device='cuda').bfloat16()
    base=t(lambda: torch.mm(dy.t(), x))
    res=[f"mm {base:.0f}us"]
    for S in (4,8,16,32,64):
        if M % S: continue
        try:
            f=lambda: torch.bmm(dy.view(S,M//S,N).transpose(1,2), x.view(S,M//S,K), out_dtype=torch.float32).sum(0)
            tt=t(f); res.append(f"S{S} {tt:.0f}")
        except Exception as e:
            res.append(f"S{S} err {str(e)[:40]}")
    # also bf16 partials
    S=16
    f=lambda: torch.bmm(dy.view(S,M//S,N).transpose(1,2), x.view(S,M//S,K)).float().sum(0)
    res.append(f"bf16part S16 {t(f):.0f}")
    ref=torch.mm(dy.t().float(), x.float()); got=torch.bmm(dy.view(16,M//16,N).transpose(1,2), x.view(16,M//16,K), out_dtype=torch.float32).sum(0)
    print((M,K,N), " | ".join(res), "relerr", ((got-ref).abs().max()/ref.abs().max()).item())
