import sys, os, torch
sys.path.insert(0, '.')
from gbx_lm_amd import ops
dev='cuda:0'
n,k,b,g = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 128
bytes_per = n*k*b//8 + 2*n*(k//g)*2
copies = max(2, int(600e6 // bytes_per) + 1)
ws=[]
for _ in range(copies):
    qw = torch.randint(0, 2**32, (n, k//32*b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n,k//g),device=dev)*0.01).to(torch.bfloat16); ze=(-sc*1.5).to(torch.bfloat16)
    ws.append((qw,sc,ze))
x = torch.randn(1,k,device=dev).to(torch.bfloat16); y = torch.empty(1,n,device=dev,dtype=torch.bfloat16)
for i in range(3*copies): ops.quantized_matmul(x,*ws[i%copies],group_size=g,bits=b,out=y)
torch.cuda.synchronize()
