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
dt, act, bits, n, k, g = torch.float16, "f16", 4, 1030, 4096, 128
gate, up = mod(n,k,bits,g,1,act), mod(n,k,bits,g,2,act)
pgu = ops.pack_scales_zeros(gate, up)
torch.manual_seed(0)
h = torch.randn(k).to(dt).to(DEV)
acc = torch.round((torch.randn(k, device=DEV)*0.7).double()*2.0**32).to(torch.int64)
nw = (1+0.1*torch.randn(k)).to(dt).to(DEV)
hsum = (h.float() + (acc.double()*2.0**-32).float().to(dt).float()).to(dt)
def run(x, acc_=None, norm=True, swiglu=True):
    o = torch.zeros(n, dtype=dt, device=DEV); xo = torch.zeros(k, dtype=dt, device=DEV)
    kw = dict(x_acc=acc_, x_out=xo) if acc_ is not None else {}
    if swiglu:
        ops.qlinear_fused(ops.make_fused_desc(x,(gate,up),(o,),norm_weight=nw if norm else None,eps=1e-5,swiglu=True,packed=pgu, **kw), x.device)
    return o
z = torch.zeros_like(acc)
print("acc=0: MODE3(h) vs MODE1(h) equal:", bool(torch.equal(run(h, z), run(h))))
print("acc: MODE3(h,acc) vs MODE1(hsum) equal:", bool(torch.equal(run(h, acc), run(hsum))))
print("no norm: MODE3(h,acc) vs MODE1(hsum) equal:", bool(torch.equal(run(h, acc, norm=False), run(hsum, norm=False))))
print("MODE3(hsum, 0) vs MODE1(hsum):", bool(torch.equal(run(hsum, z), run(hsum))))
a, b = run(h, acc), run(hsum)
bad = (a != b).nonzero().flatten()
print(len(bad), bad[:20].tolist())
