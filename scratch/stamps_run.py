import os, sys, torch, numpy as np
sys.path.insert(0, '.')
dev='cuda:0'
n,k,b,g = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 128
stamps = torch.zeros(4096*16*8, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
from gbx_lm_amd import ops
bytes_per = n*k*b//8
copies = max(2, int(600e6 // bytes_per) + 1)
ws=[]
for _ in range(copies):
    qw = torch.randint(0, 2**32, (n, k//32*b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n,k//g),device=dev)*0.01).to(torch.bfloat16); ze=(-sc*1.5).to(torch.bfloat16)
    ws.append((qw,sc,ze))
x = torch.randn(1,k,device=dev).to(torch.bfloat16); y = torch.empty(1,n,device=dev,dtype=torch.bfloat16)
for i in range(copies+3):
    stamps.zero_()
    ops.quantized_matmul(x,*ws[i%copies],group_size=g,bits=b,out=y)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1,8)
s = s[s[:,0]>0]
t0 = s[:,0].min()
rel = (s - t0)/100.0   # 100 MHz -> us
print(f"N={n} K={k} b={b}: waves={len(s)}  kernel span = {rel[:,6].max():.2f} us (first start -> last end)")
names=["start","issued(x+ring)","prologue done(barrier)","lane x ready","loop done","first drain consume","end"]
for i,nm in enumerate(names):
    c=rel[:,i]
    print(f"  {nm:26s} min {c.min():6.2f}  med {np.median(c):6.2f}  max {c.max():6.2f}")
d = rel[:,6]-rel[:,0]
print(f"  wave lifetime: med {np.median(d):.2f} max {d.max():.2f};  start spread: {rel[:,0].max():.2f}")
