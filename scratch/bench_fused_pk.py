"""Per-launch time of the four fused decode launches (2-bit 8B shapes), packed scale/zero companion vs separate tensors."""
import sys, torch
sys.path.insert(0, '.')
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dev = 'cuda:0'
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def mod(n, k, g=128):
    m = QuantizedLinear(k, n, False, g, bits)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16)
    m._buffers["scales"] = sc; m._buffers["zeros"] = (-sc * 1.5).to(torch.bfloat16)
    return m
D, I = 4096, 14336
x = torch.randn(D, device=dev).to(torch.bfloat16); nw = torch.ones(D, dtype=torch.bfloat16, device=dev)
xi = torch.randn(I, device=dev).to(torch.bfloat16)
h = torch.zeros(D, dtype=torch.bfloat16, device=dev)
L = 24
def build(kind, packed):
    out = []
    for _ in range(L):
        if kind == "qkv":
            ms = (mod(4096, D), mod(1024, D), mod(1024, D)); ys = [torch.zeros(m.output_dims, dtype=torch.bfloat16, device=dev) for m in ms]
            pk = torch.cat([ops.pack_scales_zeros(m) for m in ms]) if packed else None
            out.append((ops.make_fused_desc(x, ms, ys, norm_weight=nw, eps=1e-5, packed=pk), ms, ys, pk))
        elif kind == "o":
            ms = (mod(D, D),); pk = ops.pack_scales_zeros(ms[0]) if packed else None
            out.append((ops.make_fused_desc(x, ms, (h,), residual=h, packed=pk), ms, pk))
        elif kind == "gu":
            ms = (mod(I, D), mod(I, D)); act = torch.zeros(I, dtype=torch.bfloat16, device=dev)
            pk = ops.pack_scales_zeros(*ms) if packed else None
            out.append((ops.make_fused_desc(x, ms, (act,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pk), ms, act, pk))
        else:
            ms = (mod(D, I),); pk = ops.pack_scales_zeros(ms[0]) if packed else None
            out.append((ops.make_fused_desc(xi, ms, (h,), residual=h, packed=pk), ms, pk))
    return out
for kind in ("qkv", "o", "gu", "dn"):
    res = []
    for packed in (False, True):
        sets = build(kind, packed)
        for s in sets: ops.qlinear_fused(s[0], x.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for r in range(5):
                for s in sets: ops.qlinear_fused(s[0], x.device)
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / (5 * L))
        del sets, g
    print(f"b{bits} {kind:4s}: separate {res[0]:6.2f} us   packed {res[1]:6.2f} us")
