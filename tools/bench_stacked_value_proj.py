import sys, torch
sys.path.insert(0, '.')
import bench
bench._enable_tuned_gemms()
from grit_amd.ops.linear import weight_grad, column_sum
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
M=272000
x=torch.randn(M,512,device='cuda').bfloat16()
Ws=[torch.randn(512,512,device='cuda').bfloat16()*0.04 for _ in range(6)]; bs=[torch.randn(512,device='cuda').bfloat16() for _ in range(6)]
Wc=torch.cat(Ws); bc=torch.cat(bs)
dys=[torch.randn(M,512,device='cuda').bfloat16() for _ in range(6)]
dyc=torch.randn(M,3072,device='cuda').bfloat16()
print('fwd 6x', t(lambda: [torch.nn.functional.linear(x,w,b) for w,b in zip(Ws,bs)]), 'stacked', t(lambda: torch.nn.functional.linear(x,Wc,bc)))
def dg6():
    dx=torch.mm(dys[0],Ws[0])
    for l in range(1,6): dx.addmm_(dys[l],Ws[l])
    return dx
print('dgrad 6x', t(dg6), 'stacked', t(lambda: torch.mm(dyc,Wc)))
print('wgrad 6x', t(lambda: [weight_grad(d,x) for d in dys]), 'stacked', t(lambda: weight_grad(dyc,x)))
print('bias 6x', t(lambda: [column_sum(d,torch.bfloat16) for d in dys]), 'stacked', t(lambda: column_sum(dyc,torch.bfloat16)))
print('zero 6x', t(lambda: [torch.zeros(M,512,dtype=torch.bfloat16,device='cuda') for _ in range(6)]), 'stacked', t(lambda: torch.zeros(M,3072,dtype=torch.bfloat16,device='cuda')))
