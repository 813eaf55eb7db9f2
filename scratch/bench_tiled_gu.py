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
D,I=4096,14336; L=20
x = torch.randn(D,device=dev).to(torch.bfloat16); nw=torch.ones(D,dtype=torch.bfloat16,device=dev); act=torch.zeros(I,dtype=torch.bfloat16,device=dev)
xi = torch.randn(I,device=dev).to(torch.bfloat16); h=torch.zeros(D,dtype=torch.bfloat16,device=dev)
gu=[(mod(I,D),mod(I,D)) for _ in range(L)]
descs=[ops.make_fused_desc(x,(g,u),(act,),norm_weight=nw,eps=1e-5,swiglu=True,packed=ops.pack_scales_zeros(g,u)) for g,u in gu]
tiles=[ops.prepack_tiles(g,u) for g,u in gu]
for i in range(2): ops.qlinear_fused(descs[i],x.device); ops.gemv_tiled(x,*tiles[i],I,D,out=act,norm_weight=nw,eps=1e-5,swiglu=True)
print("gate/up fused: valu %.2f us   matrix-core %.2f us" % (timeit(lambda i: ops.qlinear_fused(descs[i%L],x.device), 3*L), timeit(lambda i: ops.gemv_tiled(x,*tiles[i%L],I,D,out=act,norm_weight=nw,eps=1e-5,swiglu=True), 3*L)))
del gu,descs,tiles
dn=[mod(D,I) for _ in range(2*L)]
descs=[ops.make_fused_desc(xi,(m,),(h,),residual=h,packed=ops.pack_scales_zeros(m)) for m in dn]
tiles=[ops.prepack_tiles(m) for m in dn]
for i in range(2): ops.qlinear_fused(descs[i],x.device); ops.gemv_tiled(xi,*tiles[i],D,I,out=h,residual=h)
print("down + residual: valu %.2f us   matrix-core %.2f us" % (timeit(lambda i: ops.qlinear_fused(descs[i%(2*L)],x.device), 4*L), timeit(lambda i: ops.gemv_tiled(xi,*tiles[i%(2*L)],D,I,out=h,residual=h), 4*L)))
