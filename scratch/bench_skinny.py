import sys, torch
sys.path.insert(0,'.')
from gbx_lm_amd import ops
dev='cuda:0'
def run(m,n,k,b,g=128,iters=50):
    copies = max(2, int(600e6 // (n*k*b//8)) + 1)
    ws=[]
    for _ in range(copies):
        qw = torch.randint(0, 2**32, (n, k//32*b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
        sc = (torch.rand((n,k//g),device=dev)*0.01).to(torch.bfloat16); ze=(-sc*1.5).to(torch.bfloat16)
        ws.append((qw,sc,ze))
    x = torch.randn(m,k,device=dev).to(torch.bfloat16); y = torch.empty(m,n,device=dev,dtype=torch.bfloat16)
    for i in range(3): ops.quantized_matmul(x,*ws[i%copies],group_size=g,bits=b,out=y)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(iters): ops.quantized_matmul(x,*ws[i%copies],group_size=g,bits=b,out=y)
    gr.replay(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/iters
    wb = n*k*b/8 + 2*n*(k//g)*2
    print(f"M={m:4d} N={n:6d} K={k:6d} b={b}: {us:8.1f} us  {2*m*n*k/us/1e6:8.1f} TFLOP/s  weights {wb/us/1e3:7.1f} GB/s")
for b in (2,4):
    for m in (8, 32, 64):
        for (n,k) in [(4096,4096),(6144,4096),(28672,4096),(4096,14336)]:
            run(m,n,k,b)
