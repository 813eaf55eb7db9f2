"""Driver for rocprofv3 --pmc passes over the SHORT decode launches (q|k|v with fused RMSNorm, o_proj + residual, down + residual),
weights cycled through > 256 MiB.  WHICH=qkv|o|down|gateup; GBX_CHAIN_FLOOR=2 gives the same launches with null weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dev = 'cuda:0'
which = os.environ.get("WHICH", "qkv")
def mod(n, k, bits=2, g=128):
    m = QuantizedLinear(k, n, False, g, bits)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16)
    m._buffers["scales"] = sc; m._buffers["zeros"] = (-sc * 1.5).to(torch.bfloat16)
    return m
k = 4096
x = torch.randn(k, device=dev).to(torch.bfloat16); nw = torch.ones(k, dtype=torch.bfloat16, device=dev)
sets = []
for _ in range(40 if which in ("qkv", "o") else 20):
    if which == "qkv":
        ms = (mod(4096, k), mod(1024, k), mod(1024, k)); ys = [torch.zeros(m.output_dims, dtype=torch.bfloat16, device=dev) for m in ms]
        pk = torch.cat([ops.pack_scales_zeros(m) for m in ms])
        sets.append((ops.make_fused_desc(x, ms, ys, norm_weight=nw, eps=1e-5, packed=pk), (ms, ys, pk)))
    elif which == "o":
        m = mod(4096, k); h = torch.zeros(4096, dtype=torch.bfloat16, device=dev); pk = ops.pack_scales_zeros(m)
        sets.append((ops.make_fused_desc(x, (m,), (h,), residual=h, packed=pk), (m, h, pk)))
    elif which == "down":
        m = mod(4096, 14336); xa = torch.randn(14336, device=dev).to(torch.bfloat16); h = torch.zeros(4096, dtype=torch.bfloat16, device=dev)
        pk = ops.pack_scales_zeros(m)
        sets.append((ops.make_fused_desc(xa, (m,), (h,), residual=h, packed=pk), (m, xa, h, pk)))
    else:
        g_, u_ = mod(14336, k), mod(14336, k); act = torch.zeros(14336, dtype=torch.bfloat16, device=dev); pk = ops.pack_scales_zeros(g_, u_)
        sets.append((ops.make_fused_desc(x, (g_, u_), (act,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pk), (g_, u_, act, pk)))
g = torch.cuda.CUDAGraph()
for d, _ in sets: ops.qlinear_fused(d, x.device)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for d, _ in sets: ops.qlinear_fused(d, x.device)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"{which}: {e0.elapsed_time(e1) * 1e3 / (10 * len(sets)):.2f} us per launch (graph replay, back to back, boundary included)")
