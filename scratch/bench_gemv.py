import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from gbx_lm_amd import ops
dev = 'cuda:0'
def mk(n,k,b,g,copies):
    out=[]
    for _ in range(copies):
        qw = torch.randint(0, 2**32, (n, k//32*b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
        sc = (torch.rand((n,k//g),device=dev)*0.01).to(torch.bfloat16); ze=(-sc*1.5).to(torch.bfloat16)
        out.append((qw,sc,ze))
    return out
def run(n,k,b,g=128,iters=200):
    bytes_per = n*k*b//8 + 2*n*(k//g)*2 + k*2 + n*2
    copies = max(2, int(600e6 // bytes_per) + 1)
    ws = mk(n,k,b,g,copies)
    x = torch.randn(1,k,device=dev).to(torch.bfloat16)
    y = torch.empty(1,n,device=dev,dtype=torch.bfloat16)
    for i in range(10): ops.quantized_matmul(x,*ws[i%copies],group_size=g,bits=b,out=y)
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        for i in range(iters): ops.quantized_matmul(x,*ws[i%copies],group_size=g,bits=b,out=y)
    g_.replay(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); g_.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/iters
    print(f"N={n:6d} K={k:6d} b={b} g={g}: {us:7.2f} us/launch  {bytes_per/us/1e3:7.1f} GB/s  ({bytes_per/1e6:.1f} MB, {copies} copies)")
for b in (2,3,4):
    for (n,k) in [(4096,4096),(6144,4096),(28672,4096),(4096,14336),(128256,4096)]:
        run(n,k,b)
