import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import gba_oracle as oracle
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
DEV='cuda:0'
def mod(n,k,bits,g,seed,act):
    d = oracle.synth_layer(n, k, bits, g, seed=seed, act_dtype=act)
    m = QuantizedLinear(k, n, False, g, bits)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m
for dt, act in ((torch.float16,"f16"),(torch.bfloat16,"bf16")):
  for bits in (4,3,2):
    for n in (1030, 2048, 14336):
        k,g=4096,128
        gate, up = mod(n,k,bits,g,1,act), mod(n,k,bits,g,2,act)
        pgu = ops.pack_scales_zeros(gate, up)
        for seed in range(3):
            torch.manual_seed(seed)
            h = torch.randn(k).to(dt).to(DEV)
            acc = torch.round((torch.randn(k, device=DEV)*0.7).double()*2.0**32).to(torch.int64)
            nw = (1+0.1*torch.randn(k)).to(dt).to(DEV)
            hsum = (h.float() + (acc.double()*2.0**-32).float().to(dt).float()).to(dt)
            a_ref = torch.zeros(n, dtype=dt, device=DEV); a_new = torch.zeros(n, dtype=dt, device=DEV); xo = torch.zeros(k, dtype=dt, device=DEV)
            ops.qlinear_fused(ops.make_fused_desc(hsum,(gate,up),(a_ref,),norm_weight=nw,eps=1e-5,swiglu=True,packed=pgu), h.device)
            ops.qlinear_fused(ops.make_fused_desc(h,(gate,up),(a_new,),norm_weight=nw,eps=1e-5,swiglu=True,packed=pgu,x_acc=acc,x_out=xo), h.device)
            a_ref2 = torch.zeros(n, dtype=dt, device=DEV)
            ops.qlinear_fused(ops.make_fused_desc(hsum,(gate,up),(a_ref2,),norm_weight=nw,eps=1e-5,swiglu=True,packed=pgu), h.device)
            bad = (a_new != a_ref).nonzero().flatten()
            print(act, bits, n, seed, "xout ok", bool(torch.equal(xo,hsum)), "ref repeat ok", bool(torch.equal(a_ref,a_ref2)), "mismatch", len(bad), bad[:6].tolist(),
                  float((a_new.float()-a_ref.float()).abs().max()))
