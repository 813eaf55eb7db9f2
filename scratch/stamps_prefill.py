import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev='cuda:0'
m,n,k,b,g = 16384,4096,4096,int(sys.argv[1]) if len(sys.argv)>1 else 4,128
stamps = torch.zeros(4096*4*4, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
from gbx_lm_amd import ops
qw = torch.randint(0, 2**32, (n, k//32*b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
sc = (torch.rand((n,k//g),device=dev)*0.01).to(torch.bfloat16); ze=(-sc*1.5).to(torch.bfloat16)
x = torch.randn(m,k,device=dev).to(torch.bfloat16); y = torch.empty(m,n,device=dev,dtype=torch.bfloat16)
for _ in range(3): ops.quantized_matmul(x,qw,sc,ze,group_size=g,bits=b,out=y)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1,4); s = s[s[:,0]>0]
print(f"waves {len(s)}; per wave cycles (median): total {np.median(s[:,0]):.0f}  load-issue {np.median(s[:,1]):.0f}  body(4 k-steps + staging) {np.median(s[:,2]):.0f}  barrier {np.median(s[:,3]):.0f}")
print(f"per chunk (64 chunks): issue {np.median(s[:,1])/64:.0f}  body {np.median(s[:,2])/64:.0f}  barrier {np.median(s[:,3])/64:.0f}   (ideal MFMA time 64 x 32 = 2048 cycles per chunk)")
