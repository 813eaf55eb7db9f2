"""In-kernel timeline of ONE decode-block GEMV launch (diagnostic -DGBX_STAMPS build, scratch/build_stamps.sh;
GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so).  usage: stamps_dg.py qkv|o|gateup|down [old|new]"""
import os, sys, torch, numpy as np
sys.path.insert(0, '.')
dev = 'cuda:0'
stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
from gbx_lm_amd import ops, _lib
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
which = sys.argv[1]; impl = sys.argv[2] if len(sys.argv) > 2 else "new"
bits = int(os.environ.get("BITS", "2")); carried = os.environ.get("CARRIED", "0") == "1"
def mod(n, k, g=128):
    m = QuantizedLinear(k, n, False, g, bits)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16)
    m._buffers["scales"] = sc; m._buffers["zeros"] = (-sc * 1.5).to(torch.bfloat16)
    return m
D, I = 4096, 14336
x = torch.randn(D, device=dev).to(torch.bfloat16); nw = torch.ones(D, dtype=torch.bfloat16, device=dev); xa = torch.randn(I, device=dev).to(torch.bfloat16)
ssq = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=dev); ssq[:256] = x.float().pow(2).view(256, -1).sum(1)
ssq_o = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=dev)
sets, keep = [], []
for _ in range(12):
    if which == "qkv":
        m = mod(6144, D); y = torch.zeros(6144, dtype=torch.bfloat16, device=dev)
        if impl == "old": pk = ops.pack_scales_zeros(m); d = ops.make_fused_desc(x, (m,), (y,), norm_weight=nw, eps=1e-5, packed=pk)
        else:
            r, wg = ops.dgemv_plan(6144, D, bits, 128, 0, True); pk = ops.dgemv_pack(m, r)
            d = ops.make_dgemv_desc(x, m, y, pk, r, norm_weight=nw, eps=1e-5, ssq_in=ssq if carried else None)
        keep.append((m, y, pk))
    elif which == "o":
        m = mod(D, D); h = torch.zeros(D, dtype=torch.bfloat16, device=dev)
        if impl == "old": pk = ops.pack_scales_zeros(m); d = ops.make_fused_desc(x, (m,), (h,), residual=h, packed=pk)
        else:
            r, wg = ops.dgemv_plan(D, D, bits, 128, 1); pk = ops.dgemv_pack(m, r)
            d = ops.make_dgemv_desc(x, m, h, pk, r, residual=h, ssq_out=ssq_o if (carried and wg > 0) else None)
        keep.append((m, h, pk))
    elif which == "down":
        m = mod(D, I); h = torch.zeros(D, dtype=torch.bfloat16, device=dev)
        if impl == "old": pk = ops.pack_scales_zeros(m); d = ops.make_fused_desc(xa, (m,), (h,), residual=h, packed=pk)
        else:
            r, wg = ops.dgemv_plan(D, I, bits, 128, 1); pk = ops.dgemv_pack(m, r)
            d = ops.make_dgemv_desc(xa, m, h, pk, r, residual=h, ssq_out=ssq_o if (carried and wg > 0) else None)
        keep.append((m, h, pk))
    else:
        g_, u_ = mod(I, D), mod(I, D); act = torch.zeros(I, dtype=torch.bfloat16, device=dev)
        if impl == "old": pk = ops.pack_scales_zeros(g_, u_); d = ops.make_fused_desc(x, (g_, u_), (act,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pk)
        else:
            r, wg = ops.dgemv_plan(I, D, bits, 128, 2, True); pk = ops.dgemv_pack(g_, r, u_)
            d = ops.make_dgemv_desc(x, g_, act, pk, r, mod_up=u_, norm_weight=nw, eps=1e-5, ssq_in=ssq if carried else None)
        keep.append((g_, u_, act, pk))
    sets.append(d)
run = (lambda d: ops.qlinear_fused(d, x.device)) if impl == "old" else (lambda d: ops.dgemv(d, x.device))
for i in range(len(sets) + 2):
    stamps.zero_()
    torch.cuda.synchronize()
    run(sets[i % len(sets)])
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 16); s = s[s[:, 0] > 0]
t0 = s[:, 0].min(); rel = (s - t0) / 100.0
rel[s == 0] = np.nan
print(f"{which} {impl} b{bits} R={os.environ.get('GBX_DG_R','-')} MIN_NW={os.environ.get('GBX_DG_MIN_NW','-')} carried={int(carried)}: waves={len(s)} span={np.nanmax(rel[:, 6]):.2f} us")
names = {0: "start", 1: "x + batch 0 issued", 8: "x landed", 9: "statistic done", 10: "image written", 2: "barrier passed", 3: "batch 1 issued, lane x ready",
         4: "batch 0 consumed", 5: "batch 1 consumed", 6: "end"}
for i in (0, 1, 8, 9, 10, 2, 3, 4, 5, 6):
    c = rel[:, i]
    if np.all(np.isnan(c)): continue
    print(f"  s{i:<2d} {names[i]:30s} min {np.nanmin(c):6.2f} med {np.nanmedian(c):6.2f} p90 {np.nanpercentile(c, 90):6.2f} max {np.nanmax(c):6.2f}")
