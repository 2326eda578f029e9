import numpy as np, torch
torch.manual_seed(0)
B,Lq,M,D,L,P=1,150,8,64,4,4
shapes=[(80,80),(40,40),(20,20),(10,10)]
starts=np.cumsum([0]+[h*w for h,w in shapes])[:-1]
S=sum(h*w for h,w in shapes)
def bf16(x): return torch.from_numpy(x.astype(np.float32)).bfloat16().float().numpy().astype(np.float64)
for spread in (0.05, 0.01):
    ref=torch.rand(B,Lq,1,1,1,2)
    loc=(ref+spread*torch.randn(B,Lq,M,L,P,2)).clamp(-0.05,1.05).numpy()
    aw=torch.softmax(torch.randn(B,Lq,M,L*P),-1).view(B,Lq,M,L,P).numpy()
    go=bf16(torch.randn(B,Lq,M,D).numpy()*0.05)
    idx_list=[];val_list=[]
    for l,(H,W) in enumerate(shapes):
        x=loc[...,l,:,0]*W-0.5; y=loc[...,l,:,1]*H-0.5   # [B,Lq,M,P]
        x0=np.floor(x).astype(int); y0=np.floor(y).astype(int)
        for dy in (0,1):
            for dx in (0,1):
                xi=x0+dx; yi=y0+dy
                w=(1-np.abs(x-xi))*(1-np.abs(y-yi))
                ok=(xi>=0)&(xi<W)&(yi>=0)&(yi<H)
                pix=starts[l]+np.clip(yi,0,H-1)*W+np.clip(xi,0,W-1)  # [B,Lq,M,P]
                m_idx=np.arange(M)[None,None,:,None]
                cell=(pix*M+m_idx)                                   # (pixel, head) cell id
                contrib=(w*aw[...,l,:]*ok)[...,None]*go[:,:,:,None,:]   # [B,Lq,M,P,D]
                idx_list.append(cell.reshape(-1)); val_list.append(contrib.reshape(-1,D))
    idx=np.concatenate(idx_list); val=np.concatenate(val_list)
    order=np.random.default_rng(0).permutation(len(idx)); idx,val=idx[order],val[order]
    truth=np.zeros((S*M,D)); np.add.at(truth,idx,val)
    fp32=np.zeros((S*M,D),np.float32); np.add.at(fp32,idx,val.astype(np.float32))
    a=bf16(fp32)                      # fp32 accumulate, one final rounding (current path)
    # bf16 accumulation: k-th occurrence rounds
    srt=np.argsort(idx,kind='stable'); si=idx[srt]; sv=val[srt]
    first=np.r_[True,si[1:]!=si[:-1]]; rank=np.arange(len(si))-np.maximum.accumulate(np.where(first,np.arange(len(si)),0))
    acc=np.zeros((S*M,D))
    for k in range(rank.max()+1):
        sel=rank==k
        acc[si[sel]]=bf16(acc[si[sel]]+bf16(sv[sel]))   # contributions themselves arrive as bf16-rounded products? keep fp32->bf16 add rounding
    cnt=np.bincount(idx,minlength=S*M)
    def rel(x): return np.linalg.norm(x-truth)/np.linalg.norm(truth)
    print("spread",spread,"updates/cell mean %.2f max %d"%(cnt[cnt>0].mean(),cnt.max()),"rel L2 err: fp32-acc+round %.2e   bf16-acc %.2e"%(rel(a),rel(acc)))
    hot=cnt>=16
    if hot.any(): print("   hot cells (>=16 updates): %d  err fp32-acc %.2e  bf16-acc %.2e"%(hot.sum(), np.linalg.norm((a-truth)[hot])/np.linalg.norm(truth[hot]), np.linalg.norm((acc-truth)[hot])/np.linalg.norm(truth[hot])))
