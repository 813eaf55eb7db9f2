import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dev='cuda:0'
def mod(n,k):
    m = QuantizedLinear(k,n,False,128,2)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k//16), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n,k//128),device=dev)*0.01).to(torch.bfloat16)
    m._buffers["scales"]=sc; m._buffers["zeros"]=(-sc*1.5).to(torch.bfloat16)
    return m
def timeit(fn, n_iter):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n_iter): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n_iter
for (n,k) in [(4096,4096),(6144,4096),(28672,4096),(4096,14336),(128256,4096)]:
    copies = max(2, int(600e6 // (n*k//4)) + 1)
    ms = [mod(n,k) for _ in range(copies)]
    tl = [ops.prepack_tiles(m) for m in ms]
    x = torch.randn(k,device=dev).to(torch.bfloat16); y = torch.empty(n,device=dev,dtype=torch.bfloat16)
    for i in range(3):
        ops.quantized_matmul(x[None],ms[i%copies].qweight,ms[i%copies].scales,ms[i%copies].zeros,group_size=128,bits=2,out=y[None])
        ops.gemv_tiled(x,*tl[i%copies],n,k,out=y)
    t_old = timeit(lambda i: ops.quantized_matmul(x[None],ms[i%copies].qweight,ms[i%copies].scales,ms[i%copies].zeros,group_size=128,bits=2,out=y[None]), 4*copies)
    t_new = timeit(lambda i: ops.gemv_tiled(x,*tl[i%copies],n,k,out=y), 4*copies)
    by = n*k//4 + n*(k//128)*4
    print(f"N={n:6d} K={k:6d}: valu path {t_old:7.2f} us ({by/t_old/1e3:6.0f} GB/s)   matrix-core path {t_new:7.2f} us ({by/t_new/1e3:6.0f} GB/s)")
    del ms, tl
