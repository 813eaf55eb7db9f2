import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dev='cuda:0'
def mod(n,k,bits,g):
    m = QuantizedLinear(k,n,False,g,bits)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k//32*bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n,k//g),device=dev)*0.01).to(torch.bfloat16)
    m._buffers["scales"]=sc; m._buffers["zeros"]=(-sc*1.5).to(torch.bfloat16)
    return m
k=4096; x=torch.randn(k,device=dev).to(torch.bfloat16); nw=torch.ones(k,dtype=torch.bfloat16,device=dev)
sets=[]
for _ in range(20):
    g_,u_=mod(14336,k,2,128),mod(14336,k,2,128); act=torch.zeros(14336,dtype=torch.bfloat16,device=dev)
    pk=ops.pack_scales_zeros(g_,u_); sets.append((ops.make_fused_desc(x,(g_,u_),(act,),norm_weight=nw,eps=1e-5,swiglu=True,packed=pk),(g_,u_,act,pk)))
for r in range(3):
    for d,_ in sets: ops.qlinear_fused(d, x.device)
torch.cuda.synchronize()
