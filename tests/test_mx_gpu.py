"""gemv_mx (csrc/gemv_mx.hip): the 2-bit decode GEMV on the block-scaled matrix cores -- packed fields read IN PLACE as FP4,
activations as four FP8 limbs with E8M0 group exponents -- against the oracle (fp64 re-expression of the GBA dequant-matmul,
quantized_linear_gba.py:195-203) at the tolerance of gemv_kernel.  Opt-in kernel: the tests run in a child process with
GBX_GEMV_MX=1."""
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
def mod(n, k, seed, act):
    d = oracle.synth_layer(n, k, 2, 128, seed=seed, act_dtype=act)
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m, d
def rms(x, w, eps):
    xf = x.float()
    return (xf * torch.rsqrt((xf * xf).mean() + eps) * w.float()).to(x.dtype)
for dt, act in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
    torch.manual_seed(7)
    # one-hot activations: every packed field meets exactly its own activation (operand slot map, scale association)
    m, d = mod(16, 1024, 1, act)
    m._buffers["scales"] = torch.ones_like(m.scales); m._buffers["zeros"] = torch.zeros_like(m.zeros)
    q = oracle.unpack(d["qweight"], 2).astype(np.float64)
    pk = ops.pack_scales_zeros(m)
    for p in list(range(0, 1024, 37)) + [15, 16, 63, 64, 127, 128, 511, 512, 1023]:
        x = torch.zeros(1024, dtype=dt, device=DEV); x[p] = 1.0
        y = torch.zeros(16, dtype=dt, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (m,), (y,), packed=pk), x.device)
        assert np.array_equal(y.float().cpu().numpy(), q[:, p]), ("one-hot", act, p)
    # random activations, every fusion of the decode block, K = 512 .. 14336
    for k, n in ((512, 16), (1024, 4096), (4096, 4096), (8192, 1024), (14336, 4096)):
        m, d = mod(n, k, 3, act)
        x = (torch.randn(k) * 1.5).to(dt).to(DEV)
        x[5] = 60.0                                   # an outlier: the group exponent follows it, the small values keep their limbs
        y = torch.zeros(n, dtype=dt, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (m,), (y,), packed=ops.pack_scales_zeros(m)), x.device)
        y64 = oracle.qlinear_ref(x.cpu()[None], d["qweight"], d["scales"], d["zeros"], 2, 128, acc=np.float64)
        oracle.assert_close(y[None], y64, act, f"plain {act} K={k}")
    k = 4096
    x = (torch.randn(k) * 1.5).to(dt).to(DEV); nw = (1 + 0.1 * torch.randn(k)).to(dt).to(DEV)
    xn = rms(x, nw, 1e-5).cpu()
    (qm, dq), (km, dk), (vm, dv) = mod(4096, k, 1, act), mod(1024, k, 2, act), mod(1024, k, 3, act)
    ys = [torch.zeros(mm.output_dims, dtype=dt, device=DEV) for mm in (qm, km, vm)]
    ops.qlinear_fused(ops.make_fused_desc(x, (qm, km, vm), ys, norm_weight=nw, eps=1e-5,
                                          packed=torch.cat([ops.pack_scales_zeros(mm) for mm in (qm, km, vm)])), x.device)
    for yy, dd, nm in zip(ys, (dq, dk, dv), "qkv"):
        oracle.assert_close(yy[None], oracle.qlinear_ref(xn[None], dd["qweight"], dd["scales"], dd["zeros"], 2, 128, acc=np.float64), act, nm)
    (gm, dg), (um, du) = mod(14336, k, 4, act), mod(14336, k, 5, act)
    o = torch.zeros(14336, dtype=dt, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(x, (gm, um), (o,), norm_weight=nw, eps=1e-5, swiglu=True, packed=ops.pack_scales_zeros(gm, um)), x.device)
    g64 = oracle.qlinear_ref(xn[None], dg["qweight"], dg["scales"], dg["zeros"], 2, 128, acc=np.float64)
    u64 = oracle.qlinear_ref(xn[None], du["qweight"], du["scales"], du["zeros"], 2, 128, acc=np.float64)
    oracle.assert_close(o[None], (g64 / (1 + np.exp(-g64))) * u64, act, "swiglu")
    (dm, dd) = mod(4096, 14336, 7, act)
    xa = torch.randn(14336).to(dt).to(DEV); h0 = torch.randn(4096).to(dt).to(DEV); h = h0.clone()
    ops.qlinear_fused(ops.make_fused_desc(xa, (dm,), (h,), residual=h, packed=ops.pack_scales_zeros(dm)), xa.device)
    y64 = oracle.qlinear_ref(xa.cpu()[None], dd["qweight"], dd["scales"], dd["zeros"], 2, 128, acc=np.float64)
    oracle.assert_close(h[None], h0.cpu().double().numpy()[None] + y64, act, "down + residual")
    # x_acc prologue: identical to the launch on the materialised sum
    hsrc = torch.randn(k).to(dt).to(DEV)
    acc = torch.round((torch.randn(k, device=DEV) * 0.7).double() * 2.0 ** 32).to(torch.int64)
    hsum = (hsrc.float() + (acc.double() * 2.0 ** -32).float().to(dt).float()).to(dt)
    pgu = ops.pack_scales_zeros(gm, um)
    a1, a2, xo = torch.zeros(14336, dtype=dt, device=DEV), torch.zeros(14336, dtype=dt, device=DEV), torch.zeros(k, dtype=dt, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(hsum, (gm, um), (a1,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pgu), x.device)
    ops.qlinear_fused(ops.make_fused_desc(hsrc, (gm, um), (a2,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pgu, x_acc=acc, x_out=xo), x.device)
    assert torch.equal(xo, hsum) and torch.equal(a1, a2)
torch.cuda.synchronize()
print("MX_OK")
'''


def test_gemv_mx_matches_oracle():
    env = dict(os.environ, GBX_GEMV_MX="1")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "MX_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
