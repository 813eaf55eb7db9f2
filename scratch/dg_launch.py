"""One kind of decode launch replayed back to back from a HIP graph (weights cycled through > 256 MiB): the previous decode GEMV
(gbx_qlinear_fused_forward, IMPL=old) against the decode-block GEMV (gbx_dgemv_forward, IMPL=new; geometry knobs GBX_DG_R,
GBX_DG_MIN_NW).  WHICH=qkv|o|gateup|down, BITS=2|3|4, CARRIED=1 feeds the norm'd launches a producer's partial statistic.
GBX_CHAIN_FLOOR=2: the same launches with null weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, _lib
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dev = 'cuda:0'
which = os.environ.get("WHICH", "qkv")
impl = os.environ.get("IMPL", "new")
bits = int(os.environ.get("BITS", "2"))
carried = os.environ.get("CARRIED", "0") == "1"
D, I = int(os.environ.get("HIDDEN", "4096")), int(os.environ.get("INTER", "14336"))
NQKV = int(os.environ.get("NQKV", "6144"))
def mod(n, k, g=128):
    m = QuantizedLinear(k, n, False, g, bits)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16)
    m._buffers["scales"] = sc; m._buffers["zeros"] = (-sc * 1.5).to(torch.bfloat16)
    return m
x = torch.randn(D, device=dev).to(torch.bfloat16); nw = torch.ones(D, dtype=torch.bfloat16, device=dev)
xa = torch.randn(I, device=dev).to(torch.bfloat16)
ssq = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=dev); ssq[:256] = x.float().pow(2).view(256, -1).sum(1)
ssq_o = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=dev)
sets, keep = [], []
nsets = 40 if which in ("qkv", "o") else 20
for _ in range(nsets):
    if which == "qkv":
        m = mod(NQKV, D); y = torch.zeros(NQKV, dtype=torch.bfloat16, device=dev)
        if impl == "old":
            pk = ops.pack_scales_zeros(m); d = ops.make_fused_desc(x, (m,), (y,), norm_weight=nw, eps=1e-5, packed=pk)
        else:
            r, wg = ops.dgemv_plan(NQKV, D, bits, 128, 0, True); pk = ops.dgemv_pack(m, r)
            d = ops.make_dgemv_desc(x, m, y, pk, r, norm_weight=nw, eps=1e-5, ssq_in=ssq if carried else None)
        keep.append((m, y, pk))
    elif which == "o":
        m = mod(D, D); h = torch.zeros(D, dtype=torch.bfloat16, device=dev)
        if impl == "old":
            pk = ops.pack_scales_zeros(m); d = ops.make_fused_desc(x, (m,), (h,), residual=h, packed=pk)
        else:
            r, wg = ops.dgemv_plan(D, D, bits, 128, 1); pk = ops.dgemv_pack(m, r)
            d = ops.make_dgemv_desc(x, m, h, pk, r, residual=h, ssq_out=ssq_o if (carried and wg > 0) else None)
        keep.append((m, h, pk))
    elif which == "down":
        m = mod(D, I); h = torch.zeros(D, dtype=torch.bfloat16, device=dev)
        if impl == "old":
            pk = ops.pack_scales_zeros(m); d = ops.make_fused_desc(xa, (m,), (h,), residual=h, packed=pk)
        else:
            r, wg = ops.dgemv_plan(D, I, bits, 128, 1); pk = ops.dgemv_pack(m, r)
            d = ops.make_dgemv_desc(xa, m, h, pk, r, residual=h, ssq_out=ssq_o if (carried and wg > 0) else None)
        keep.append((m, h, pk))
    else:
        g_, u_ = mod(I, D), mod(I, D); act = torch.zeros(I, dtype=torch.bfloat16, device=dev)
        if impl == "old":
            pk = ops.pack_scales_zeros(g_, u_); d = ops.make_fused_desc(x, (g_, u_), (act,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pk)
        else:
            r, wg = ops.dgemv_plan(I, D, bits, 128, 2, True); pk = ops.dgemv_pack(g_, r, u_)
            d = ops.make_dgemv_desc(x, g_, act, pk, r, mod_up=u_, norm_weight=nw, eps=1e-5, ssq_in=ssq if carried else None)
        keep.append((g_, u_, act, pk))
    sets.append(d)
run = (lambda d: ops.qlinear_fused(d, x.device)) if impl == "old" else (lambda d: ops.dgemv(d, x.device))
# never time a broken kernel: the first weight set through the other implementation too, outputs compared
def check():
    kp = keep[0]
    out = kp[-2] if which != "gateup" else kp[2]
    if which in ("o", "down"): out.zero_()
    run(sets[0]); torch.cuda.synchronize(); got = out.clone()
    if which in ("o", "down"): out.zero_()
    xin = xa if which == "down" else x
    if impl == "new":
        if which == "gateup": d2 = ops.make_fused_desc(x, (kp[0], kp[1]), (out,), norm_weight=nw, eps=1e-5, swiglu=True, packed=ops.pack_scales_zeros(kp[0], kp[1]))
        elif which == "qkv": d2 = ops.make_fused_desc(x, (kp[0],), (out,), norm_weight=nw, eps=1e-5, packed=ops.pack_scales_zeros(kp[0]))
        else: d2 = ops.make_fused_desc(xin, (kp[0],), (out,), residual=out, packed=ops.pack_scales_zeros(kp[0]))
        ops.qlinear_fused(d2, x.device); torch.cuda.synchronize()
        same = (got == out).float().mean().item()
        err = (got.float() - out.float()).abs().max().item() / (out.float().abs().max().item() + 1e-30)
        if os.environ.get("GBX_CHAIN_FLOOR") is None:
            assert same > 0.97 and err < 2e-2, f"{which}: new vs old outputs differ: equal {same:.4f}, max rel err {err:.3g}"
        return f"equal {same:.4f}"
    return ""
chk = check()
g = torch.cuda.CUDAGraph()
for d in sets: run(d)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for d in sets: run(d)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / (10 * len(sets)))
tag = f"R={os.environ.get('GBX_DG_R','-')} MIN_NW={os.environ.get('GBX_DG_MIN_NW','-')} carried={int(carried)}" if impl == "new" else ""
print(f"{which:7s} b{bits} {impl} {tag}: {best:.2f} us per launch  {chk}")
