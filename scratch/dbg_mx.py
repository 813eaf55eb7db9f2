import sys, numpy as np, torch
sys.path.insert(0, '.')
DEV = 'cuda:0'
import importlib.util
spec = importlib.util.spec_from_file_location("oracle_mod", "oracle/gba_oracle.py"); oracle = importlib.util.module_from_spec(spec); spec.loader.exec_module(oracle)
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
def mod(n, k, seed, dt):
    act = "bf16" if dt == torch.bfloat16 else "f16"
    d = oracle.synth_layer(n, k, 2, 128, seed=seed, act_dtype=act)
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m, d
def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-9))
for dt in (torch.bfloat16, torch.float16):
    torch.manual_seed(1)
    k = 4096
    x = (torch.randn(k) * 1.5).to(dt).to(DEV); nw = (1 + 0.1 * torch.randn(k)).to(dt).to(DEV)
    # q/k/v with norm
    (q, dq), (kk, dk), (v, dv) = mod(4096, k, 1, dt), mod(1024, k, 2, dt), mod(1024, k, 3, dt)
    def run(packed):
        ys = [torch.zeros(m.output_dims, dtype=dt, device=DEV) for m in (q, kk, v)]
        ops.qlinear_fused(ops.make_fused_desc(x, (q, kk, v), ys, norm_weight=nw, eps=1e-5, packed=packed), x.device); return ys
    ref = run(None); new = run(torch.cat([ops.pack_scales_zeros(m) for m in (q, kk, v)]))
    print(dt, "qkv  ", [rel(a, b) for a, b in zip(new, ref)], [float((a != b).float().mean()) for a, b in zip(new, ref)])
    # gate/up swiglu
    (g, dg), (u, du) = mod(14336, k, 4, dt), mod(14336, k, 5, dt)
    def rung(packed):
        o = torch.zeros(14336, dtype=dt, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (g, u), (o,), norm_weight=nw, eps=1e-5, swiglu=True, packed=packed), x.device); return o
    ref = rung(None); new = rung(ops.pack_scales_zeros(g, u))
    print(dt, "gateup", rel(new, ref), float((new != ref).float().mean()))
    # o with residual
    (o_, do) = mod(4096, k, 6, dt); h0 = torch.randn(4096).to(dt).to(DEV)
    def runo(packed):
        h = h0.clone(); ops.qlinear_fused(ops.make_fused_desc(x, (o_,), (h,), residual=h, packed=packed), x.device); return h
    ref = runo(None); new = runo(ops.pack_scales_zeros(o_))
    print(dt, "o     ", rel(new, ref), float((new != ref).float().mean()))
    # down K = 14336
    (d_, dd) = mod(4096, 14336, 7, dt); xa = torch.randn(14336).to(dt).to(DEV)
    def rund(packed):
        h = h0.clone(); ops.qlinear_fused(ops.make_fused_desc(xa, (d_,), (h,), residual=h, packed=packed), xa.device); return h
    ref = rund(None); new = rund(ops.pack_scales_zeros(d_))
    print(dt, "down  ", rel(new, ref), float((new != ref).float().mean()))
    # oracle check for o (no residual effect): y64
    y = torch.zeros(4096, dtype=dt, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(x, (o_,), (y,), packed=ops.pack_scales_zeros(o_)), x.device)
    y64 = oracle.qlinear_ref(x.cpu()[None], do["qweight"], do["scales"], do["zeros"], 2, 128, acc=np.float64)[0]
    yl = torch.zeros(4096, dtype=dt, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(x, (o_,), (yl,)), x.device)
    e_new = np.abs(y.cpu().double().numpy() - y64).max(); e_old = np.abs(yl.cpu().double().numpy() - y64).max()
    print(dt, "oracle max abs err  mx %.5f  legacy %.5f  (|y| max %.3f)" % (e_new, e_old, np.abs(y64).max()))
