import sys, torch
sys.path.insert(0,'.')
from gbx_lm_amd import ops
dev='cuda:0'
def run(m,n,k,b,g=128,iters=20):
    qw = torch.randint(0, 2**32, (n, k//32*b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n,k//g),device=dev)*0.01).to(torch.bfloat16); ze=(-sc*1.5).to(torch.bfloat16)
    x = torch.randn(m,k,device=dev).to(torch.bfloat16); y = torch.empty(m,n,device=dev,dtype=torch.bfloat16)
    for _ in range(3): ops.quantized_matmul(x,qw,sc,ze,group_size=g,bits=b,out=y)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.quantized_matmul(x,qw,sc,ze,group_size=g,bits=b,out=y)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/iters
    print(f"M={m:6d} N={n:6d} K={k:6d} b={b}: {ms*1e3:9.1f} us  {2*m*n*k/ms/1e9:8.1f} TFLOP/s  weights {n*k*b/8/ms/1e6:7.1f} GB/s")
for b in (2,4):
    for m in (32, 256, 4096, 32768):
        run(m, 4096, 4096, b)
    run(32, 28672, 4096, b); run(4096, 14336, 4096, b)
