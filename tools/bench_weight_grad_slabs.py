import sys, torch
sys.path.insert(0, '.')
from grit_amd.ops.linear import slab_sum
def t(fn, it=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
shapes = [(51200,512,512),(51200,512,1536),(51200,512,2048),(51200,2048,512),(204800,256,256),(204800,256,768),(204800,256,1024),(204800,1024,256),
          (12800,1024,1024),(12800,1024,3072),(12800,1024,4096),(12800,4096,1024),(272000,512,512)]
for (M,K,N) in shapes:
    x=torch.randn(M,K,device='cuda').bfloat16(); dy=torch.randn(M,N,device='cuda').bfloat16()
    res=[f"mm {t(lambda: torch.mm(dy.t(), x)):.0f}"]
    for S in (2,4,5,8,10,16,20,25,32,40,50,64,100,128):
        if M % S or M // S < 400: continue
        f=lambda: slab_sum(torch.bmm(dy.view(S,M//S,N).transpose(1,2), x.view(S,M//S,K), out_dtype=torch.float32).unsqueeze(0), torch.bfloat16)
        res.append(f"S{S}:{t(f):.0f}")
    print((M,K,N), " ".join(res), flush=True)
