"""gemv_mxd (csrc/gemv_mxd.hip): the 2-bit decode GEMV with ONE packed row per scaled-MFMA operand (16 consecutive groups as the 16
columns, diagonal of the result) -- packed fields read in place as FP4, bf16 activations as two exact FP8 limbs with E8M0 exponents per
32 values -- against the oracle (fp64 re-expression of the GBA dequant-matmul, quantized_linear_gba.py:195-203) at the tolerance of
gemv_kernel.  Opt-in kernel: the tests run in a child process with GBX_GEMV_MXD=1."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.experimental]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from oracle import gba_oracle as oracle
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
DEV = "cuda:0"
dt, act = torch.bfloat16, "bf16"
def mod(n, k, seed):
    d = oracle.synth_layer(n, k, 2, 128, seed=seed, act_dtype=act)
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m, d
def rms(x, w, eps):
    xf = x.float()
    return (xf * torch.rsqrt((xf * xf).mean() + eps) * w.float()).to(x.dtype)
torch.manual_seed(11)
# one-hot activations: every packed field meets exactly its own activation (operand slot map, block scales, diagonal pick)
for k in (2048, 4096):
    m, d = mod(16, k, 1)
    m._buffers["scales"] = torch.ones_like(m.scales); m._buffers["zeros"] = torch.zeros_like(m.zeros)
    q = oracle.unpack(d["qweight"], 2).astype(np.float64)
    pk = ops.pack_scales_zeros(m)
    for p in list(range(0, k, 61)) + [15, 16, 31, 32, 63, 64, 127, 128, 2047, k - 1]:
        x = torch.zeros(k, dtype=dt, device=DEV); x[p] = 3.0
        y = torch.zeros(16, dtype=dt, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (m,), (y,), packed=pk), x.device)
        assert np.array_equal(y.float().cpu().numpy(), 3.0 * q[:, p]), ("one-hot", k, p)
# exactness of the two limbs: any bf16 activations with a narrow dynamic range inside a 32-value block reproduce the fp32 sums of
# exact products (integer weights, scale 1, zero 0): compare with the integer dot in fp64
m, d = mod(64, 4096, 2)
m._buffers["scales"] = torch.ones_like(m.scales); m._buffers["zeros"] = torch.zeros_like(m.zeros)
q = oracle.unpack(d["qweight"], 2).astype(np.float64)
x = (torch.randint(128, 256, (4096,)).float() / 128.0 * torch.where(torch.rand(4096) < 0.5, -1.0, 1.0)).to(dt).to(DEV)   # 8 significant bits each
y = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
ops.qlinear_fused(ops.make_fused_desc(x, (m,), (y,), packed=ops.pack_scales_zeros(m)), x.device)
want = torch.from_numpy(q @ x.double().cpu().numpy()).to(dt)
assert torch.equal(y.cpu(), want), float((y.cpu().float() - want.float()).abs().max())
# random activations with outliers, every fusion of the decode block at K = 2048 / 4096, both geometries
for k, n in ((2048, 512), (4096, 4096), (4096, 130), (2048, 14336)):
    mm, dd = mod(n, k, 3)
    x = (torch.randn(k) * 1.5).to(dt).to(DEV)
    x[5] = 60.0; x[77] = -1e-3
    y = torch.zeros(n, dtype=dt, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(x, (mm,), (y,), packed=ops.pack_scales_zeros(mm)), x.device)
    y64 = oracle.qlinear_ref(x.cpu()[None], dd["qweight"], dd["scales"], dd["zeros"], 2, 128, acc=np.float64)
    oracle.assert_close(y[None], y64, act, f"plain K={k} N={n}")
k = 4096
x = (torch.randn(k) * 1.5).to(dt).to(DEV); nw = (1 + 0.1 * torch.randn(k)).to(dt).to(DEV)
xn = rms(x, nw, 1e-5).cpu()
(qm, dq), (km, dk), (vm, dv) = mod(4096, k, 1), mod(1024, k, 2), mod(1024, k, 3)
ys = [torch.zeros(mm.output_dims, dtype=dt, device=DEV) for mm in (qm, km, vm)]
ops.qlinear_fused(ops.make_fused_desc(x, (qm, km, vm), ys, norm_weight=nw, eps=1e-5,
                                      packed=torch.cat([ops.pack_scales_zeros(mm) for mm in (qm, km, vm)])), x.device)
for yy, dd, nm in zip(ys, (dq, dk, dv), "qkv"):
    oracle.assert_close(yy[None], oracle.qlinear_ref(xn[None], dd["qweight"], dd["scales"], dd["zeros"], 2, 128, acc=np.float64), act, nm)
(gm, dg), (um, du) = mod(14336, k, 4), mod(14336, k, 5)
o = torch.zeros(14336, dtype=dt, device=DEV)
ops.qlinear_fused(ops.make_fused_desc(x, (gm, um), (o,), norm_weight=nw, eps=1e-5, swiglu=True, packed=ops.pack_scales_zeros(gm, um)), x.device)
g64 = oracle.qlinear_ref(xn[None], dg["qweight"], dg["scales"], dg["zeros"], 2, 128, acc=np.float64)
u64 = oracle.qlinear_ref(xn[None], du["qweight"], du["scales"], du["zeros"], 2, 128, acc=np.float64)
oracle.assert_close(o[None], (g64 / (1 + np.exp(-g64))) * u64, act, "swiglu")
(om, do) = mod(4096, k, 7)
xa = torch.randn(k).to(dt).to(DEV); h0 = torch.randn(4096).to(dt).to(DEV); h = h0.clone()
ops.qlinear_fused(ops.make_fused_desc(xa, (om,), (h,), residual=h, packed=ops.pack_scales_zeros(om)), xa.device)
y64 = oracle.qlinear_ref(xa.cpu()[None], do["qweight"], do["scales"], do["zeros"], 2, 128, acc=np.float64)
oracle.assert_close(h[None], h0.cpu().double().numpy()[None] + y64, act, "o + residual")
torch.cuda.synchronize()
print("MXD_OK")
'''


def test_gemv_mxd_matches_oracle():
    env = dict(os.environ, GBX_GEMV_MXD="1")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "MXD_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
