import os, sys, torch, numpy as np
sys.path.insert(0, '.')
dev='cuda:0'
stamps = torch.zeros(4096*16*8, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
def mod(n,k,bits,g):
    m = QuantizedLinear(k,n,False,g,bits)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k//32*bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n,k//g),device=dev)*0.01).to(torch.bfloat16)
    m._buffers["scales"]=sc; m._buffers["zeros"]=(-sc*1.5).to(torch.bfloat16)
    return m
which = sys.argv[1]
k=4096; x=torch.randn(k,device=dev).to(torch.bfloat16); nw=torch.ones(k,dtype=torch.bfloat16,device=dev)
sets=[]
for _ in range(12):
    if which=="gateup":
        g_,u_=mod(14336,k,2,128),mod(14336,k,2,128); act=torch.zeros(14336,dtype=torch.bfloat16,device=dev)
        sets.append((ops.make_fused_desc(x,(g_,u_),(act,),norm_weight=nw,eps=1e-5,swiglu=True),(g_,u_,act)))
    elif which=="qkv":
        q,kk,v=mod(4096,k,2,128),mod(1024,k,2,128),mod(1024,k,2,128); ys=[torch.zeros(m.output_dims,dtype=torch.bfloat16,device=dev) for m in (q,kk,v)]
        sets.append((ops.make_fused_desc(x,(q,kk,v),ys,norm_weight=nw,eps=1e-5),(q,kk,v,ys)))
    elif which=="down":
        d=mod(4096,14336,2,128); xa=torch.randn(14336,device=dev).to(torch.bfloat16); h=torch.zeros(4096,dtype=torch.bfloat16,device=dev)
        sets.append((ops.make_fused_desc(xa,(d,),(h,),residual=h),(d,xa,h)))
    elif which=="o":
        d=mod(4096,4096,2,128); h=torch.zeros(4096,dtype=torch.bfloat16,device=dev)
        sets.append((ops.make_fused_desc(x,(d,),(h,),residual=h),(d,h)))
for i in range(len(sets)+2):
    stamps.zero_()
    ops.qlinear_fused(sets[i%len(sets)][0], x.device)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1,8); s = s[s[:,0]>0]
t0 = s[:,0].min(); rel=(s-t0)/100.0
print(f"{which}: waves={len(s)} span={rel[:,6].max():.2f} us")
for i,nm in enumerate(["start","x+first batch issued","prologue done","lane x ready","ring issued/loop done","first drain consume","end"]):
    c=rel[:,i]; print(f"  {nm:24s} min {c.min():6.2f} med {np.median(c):6.2f} max {c.max():6.2f}")
# --- tail analysis: per-workgroup end time vs block id / start time
nwav = 16
blk_end = rel[:,6].reshape(-1,nwav).max(1) if len(rel)%nwav==0 else None
if blk_end is not None:
    blk_start = rel[:,0].reshape(-1,nwav).min(1)
    blk_pro = rel[:,2].reshape(-1,nwav).max(1)
    order = np.argsort(blk_end)
    print("  per-WG end: min %.2f p25 %.2f med %.2f p75 %.2f p90 %.2f max %.2f" % (blk_end.min(), np.percentile(blk_end,25), np.median(blk_end), np.percentile(blk_end,75), np.percentile(blk_end,90), blk_end.max()))
    print("  slowest 8 WGs: ids", order[-8:], "start", np.round(blk_start[order[-8:]],2), "prologue", np.round(blk_pro[order[-8:]],2), "end", np.round(blk_end[order[-8:]],2))
    print("  fastest 8 WGs: ids", order[:8], "start", np.round(blk_start[order[:8]],2), "prologue", np.round(blk_pro[order[:8]],2), "end", np.round(blk_end[order[:8]],2))
    print("  corr(end, id%8)=", [round(float(blk_end[np.arange(len(blk_end))%8==x].mean()),2) for x in range(8)])
    # per-wave durations of loop phases
    for a_,b_,nm in ((3,4,"loop"),(4,6,"drain")):
        dd = rel[:,b_]-rel[:,a_]
        print(f"  {nm}: min {dd.min():.2f} med {np.median(dd):.2f} p90 {np.percentile(dd,90):.2f} max {dd.max():.2f}")
