"""GPU: the decode-block GEMV (gbx_dgemv_forward, csrc/dgemv_impl.h) through the C-ABI against the oracle
(fp64 x . dequantised weights; gbx_lm/models/quantized_linear_gba.py:195-205 restated in oracle/gba_oracle.py), against the
previous decode kernel (gbx_qlinear_fused_forward: bit-identical where the arithmetic order is the same), with the carried RMSNorm
statistic, and at the exact launch geometries the bench times (Llama-3-8B / 3.2-3B shapes, every bit width)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mk(oracle, n, k, bits, g, seed):
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    d = oracle.synth_layer(n, k, bits, g, seed=seed)
    m = QuantizedLinear(k, n, False, g, bits)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m, d


def _rms(x, w, eps):
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()).to(x.dtype)


def _ref(oracle, x_cpu, d, bits, g):
    return oracle.qlinear_ref(x_cpu[None], d["qweight"], d["scales"], d["zeros"], bits, g, acc=np.float64)


def _plan(ops, m, epi, norm=False):
    pl = ops.dgemv_plan(m.output_dims, m.input_dims, m.bits, m.group_size, epi, norm)
    assert pl is not None, "dgemv must cover this shape"
    return pl


def _ulps(a, b):
    """distance in bf16 / fp16 units of the larger magnitude, elementwise"""
    a, b = a.float(), b.float()
    scale = torch.maximum(a.abs(), b.abs()).clamp_min(1e-30)
    return ((a - b).abs() / scale * 256).max().item()


# (n, k): one lane pass (P1 KS1), two passes (P2 KS1, 3/4 bits at K 4096), K split over four waves (KS4), both
SHAPES = [(256, 512), (520, 4096), (1024, 8192), (136, 14336)]


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("n,k", SHAPES)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_dgemv_plain_and_residual_vs_oracle(oracle, bits, n, k, dt):
    from gbx_lm_amd import ops, _lib
    g = 128 if (n + k) % 3 else 64
    m, d = _mk(oracle, n, k, bits, g, seed=bits * 7 + n)
    if dt == torch.float16:
        for nm in ("scales", "zeros"):
            m._buffers[nm] = m._buffers[nm].to(dt)
        d = dict(d, scales=m.scales.cpu(), zeros=m.zeros.cpu())
    x = (torch.randn(k) * 2).to(dt).to(DEV)
    y64 = _ref(oracle, x.cpu(), d, bits, g)
    name = "bf16" if dt == torch.bfloat16 else "fp16"
    # plain
    r, wg = _plan(ops, m, _lib.GBX_DG_PLAIN)
    pk = ops.dgemv_pack(m, r)
    y = torch.full((n,), 7.0, dtype=dt, device=DEV)
    ops.dgemv(ops.make_dgemv_desc(x, m, y, pk, r), x.device)
    oracle.assert_close(y[None], y64, name, f"dgemv plain b{bits} {n}x{k}")
    # residual, in place, with the partial statistic of the new stream
    h = torch.randn(n).to(dt).to(DEV)
    h0 = h.clone()
    r1, wg = _plan(ops, m, _lib.GBX_DG_RESIDUAL)
    pk1 = pk if r1 == r else ops.dgemv_pack(m, r1)
    ssq = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=DEV) if wg > 0 else None
    ops.dgemv(ops.make_dgemv_desc(x, m, h, pk1, r1, residual=h, ssq_out=ssq), x.device)
    oracle.assert_close(h[None], h0.cpu().double().numpy()[None] + y64, name, "dgemv residual")
    if ssq is not None:
        want = h.double().pow(2).sum().item()
        got = ssq.double().sum().item()
        assert abs(got - want) <= 1e-5 * want, (got, want)
        assert (ssq[wg:] == 0).all(), "the launch writes exactly the partials the plan reports"
    # the previous kernel on the same inputs: same arithmetic, same order -> same bits
    y2 = torch.zeros(n, dtype=dt, device=DEV)
    if n % 2 == 0:
        ops.qlinear_fused(ops.make_fused_desc(x, (m,), (y2,), packed=ops.pack_scales_zeros(m)), x.device)
        assert torch.equal(y, y2), f"dgemv vs gemv_kernel: {(y != y2).sum().item()} of {n} differ, {_ulps(y, y2):.2f} ulp"


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("n,k,g", [(512, 4096, 128), (1032, 4096, 64), (384, 1024, 128), (256, 3072, 128)])
@pytest.mark.parametrize("carried", [False, True])
def test_dgemv_norm_swiglu_vs_oracle(oracle, bits, n, k, g, carried):
    from gbx_lm_amd import ops, _lib
    gate, dg = _mk(oracle, n, k, bits, g, seed=1 + bits)
    up, du = _mk(oracle, n, k, bits, g, seed=2 + bits)
    x = (torch.randn(k) * 2).to(torch.bfloat16).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(torch.bfloat16).to(DEV)
    act = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    r, _ = _plan(ops, gate, _lib.GBX_DG_SWIGLU, True)
    pk = ops.dgemv_pack(gate, r, up)
    ssq = None
    if carried:      # partial sums as a producer leaves them: any split of sum(x^2) over the slots
        sq = x.float().pow(2)
        ssq = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=DEV)
        parts = sq.view(256, -1).sum(1)
        ssq[:256] = parts
    ops.dgemv(ops.make_dgemv_desc(x, gate, act, pk, r, mod_up=up, norm_weight=nw, eps=1e-5, ssq_in=ssq), x.device)
    xn = _rms(x, nw, 1e-5).cpu()
    g64, u64 = _ref(oracle, xn, dg, bits, g), _ref(oracle, xn, du, bits, g)
    oracle.assert_close(act[None], (g64 / (1 + np.exp(-g64))) * u64, "bf16", f"dgemv swiglu b{bits} carried={carried}")
    if not carried:
        act2 = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (gate, up), (act2,), norm_weight=nw, eps=1e-5, swiglu=True,
                                              packed=ops.pack_scales_zeros(gate, up)), x.device)
        same = (act == act2).float().mean().item()
        # equal workgroup width -> equal statistic order -> equal bits; a different width may move rstd by an ulp
        assert same > 0.98 and _ulps(act, act2) <= 4.0, (same, _ulps(act, act2))


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_dgemv_norm_plain_qkv(oracle, bits):
    from gbx_lm_amd import ops, _lib
    k, n, g = 4096, 768, 128
    m, d = _mk(oracle, n, k, bits, g, seed=5)
    x = (torch.randn(k) * 3).to(torch.bfloat16).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(torch.bfloat16).to(DEV)
    y = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    r, _ = _plan(ops, m, _lib.GBX_DG_PLAIN, True)
    ops.dgemv(ops.make_dgemv_desc(x, m, y, ops.dgemv_pack(m, r), r, norm_weight=nw, eps=1e-5), x.device)
    oracle.assert_close(y[None], _ref(oracle, _rms(x, nw, 1e-5).cpu(), d, bits, g), "bf16", "dgemv norm + plain")


@pytest.mark.parametrize("r", [2, 4])
@pytest.mark.parametrize("epi", ["plain", "residual", "swiglu"])
def test_dgemv_either_packing(oracle, r, epi):
    """gbx_dgemv_forward takes the companion packed for 2 or for 4 rows per batch wherever the rows divide; same results."""
    from gbx_lm_amd import ops
    n, k, g, bits = 512, 4096, 128, 2
    m, d = _mk(oracle, n, k, bits, g, seed=3)
    up, du = _mk(oracle, n, k, bits, g, seed=4)
    x = (torch.randn(k) * 2).to(torch.bfloat16).to(DEV)
    y64 = _ref(oracle, x.cpu(), d, bits, g)
    y = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    if epi == "swiglu":
        u64 = _ref(oracle, x.cpu(), du, bits, g)
        ops.dgemv(ops.make_dgemv_desc(x, m, y, ops.dgemv_pack(m, r, up), r, mod_up=up), x.device)
        oracle.assert_close(y[None], (y64 / (1 + np.exp(-y64))) * u64, "bf16", f"swiglu R={r}")
    elif epi == "residual":
        h = torch.randn(n).to(torch.bfloat16).to(DEV)
        h0 = h.clone()
        ops.dgemv(ops.make_dgemv_desc(x, m, h, ops.dgemv_pack(m, r), r, residual=h), x.device)
        oracle.assert_close(h[None], h0.cpu().double().numpy()[None] + y64, "bf16", f"residual R={r}")
    else:
        ops.dgemv(ops.make_dgemv_desc(x, m, y, ops.dgemv_pack(m, r), r), x.device)
        oracle.assert_close(y[None], y64, "bf16", f"plain R={r}")


def test_dgemv_pack_layout():
    """gbx_dgemv_pack against the layout restated with torch ops: [batch][group][row of the batch]."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    n, k, g = 64, 1024, 128
    mods = []
    for s in range(2):
        m = QuantizedLinear(k, n, False, g, 2)
        m._buffers["scales"] = torch.randn(n, k // g).to(torch.bfloat16).to(DEV)
        m._buffers["zeros"] = torch.randn(n, k // g).to(torch.bfloat16).to(DEV)
        mods.append(m)
    word = lambda m: (m.scales.view(torch.int16).to(torch.int32) & 0xFFFF) | (m.zeros.view(torch.int16).to(torch.int32) << 16)
    for r in (2, 4):
        got = ops.dgemv_pack(mods[0], r)
        want = word(mods[0]).view(n // r, r, k // g).permute(0, 2, 1).contiguous()
        assert torch.equal(got, want)
        got = ops.dgemv_pack(mods[0], r, mods[1])
        h = r // 2
        wg, wu = word(mods[0]).view(n // h, h, k // g), word(mods[1]).view(n // h, h, k // g)
        want = torch.cat([wg, wu], 1).permute(0, 2, 1).contiguous()
        assert torch.equal(got, want)


def test_dgemv_rejections():
    from gbx_lm_amd import ops, _lib
    assert ops.dgemv_plan(4096, 4096, 2, 32, 0) is None          # two groups per lane: gemv_kernel's split form
    assert ops.dgemv_plan(4097, 4096, 2, 128, 0) is None         # rows do not divide
    assert ops.dgemv_plan(4096, 14336, 2, 128, 2) is None        # SwiGLU needs whole rows per wave
    assert ops.dgemv_plan(4096, 4096 * 16, 4, 128, 0) is None    # beyond 512 lane slots
    d = _lib.DgDesc()
    with pytest.raises(ValueError):
        ops.dgemv(d, torch.device(DEV))                          # struct_size 0


# ---- the launches the bench times, at their own geometry (Llama-3-8B and 3.2-3B shapes) ----------------------
FULL = [  # (name, N, K, epilogue)
    ("8B qkv", 6144, 4096, "plain+norm"), ("8B o", 4096, 4096, "residual"), ("8B gate|up", 14336, 4096, "swiglu+norm"),
    ("8B down", 4096, 14336, "residual"),
    ("3B qkv", 5120, 3072, "plain+norm"), ("3B o", 3072, 3072, "residual"), ("3B gate|up", 8192, 3072, "swiglu+norm"),
    ("3B down", 3072, 8192, "residual"),
]


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("name,n,k,kind", FULL)
def test_dgemv_bench_geometry_vs_dequantised_fp64(oracle, bits, name, n, k, kind):
    """The exact instantiation / grid the decode runner launches, against fp64 x . gbx_dequantize(weights) (the GPU dequantise
    is itself bit-exact against the C oracle, tests/test_fullsize_gpu.py) -- the full-size rows make the numpy oracle too slow."""
    from gbx_lm_amd import ops, _lib
    g = 128
    torch.manual_seed(bits * 100 + n % 97)

    def rand_mod(seed):
        from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
        m = QuantizedLinear(k, n, False, g, bits)
        gen = torch.Generator().manual_seed(seed)
        m._buffers["qweight"] = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, k // 32 * bits), dtype=torch.int32, generator=gen).view(torch.uint32).to(DEV)
        sc = (0.5 + torch.rand(n, k // g, generator=gen)) * (2.0 / (2 ** bits - 1)) * 0.02
        m._buffers["scales"] = sc.to(torch.bfloat16).to(DEV)
        m._buffers["zeros"] = (-m.scales.float().cpu() * (2 ** bits - 1) / 2 + 0.1 * sc * torch.randn(n, k // g, generator=gen)).to(torch.bfloat16).to(DEV)
        return m

    deq = lambda m: ops.dequantize(m.qweight, m.scales, m.zeros, m.group_size, m.bits, dtype=torch.float32).double()
    x = (torch.randn(k) * (1.0 if "norm" in kind else 0.5)).to(torch.bfloat16).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(torch.bfloat16).to(DEV)
    m = rand_mod(1)
    if kind == "swiglu+norm":
        up = rand_mod(2)
        r, wg = _plan(ops, m, _lib.GBX_DG_SWIGLU, True)
        act = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        ops.dgemv(ops.make_dgemv_desc(x, m, act, ops.dgemv_pack(m, r, up), r, mod_up=up, norm_weight=nw, eps=1e-5), x.device)
        xn = _rms(x, nw, 1e-5).double()
        g64, u64 = deq(m) @ xn, deq(up) @ xn
        # the op-by-op roundings of qllama.py:115 are part of the contract: compare with them applied
        gb, ub = g64.to(torch.bfloat16).double(), u64.to(torch.bfloat16).double()
        want = (gb / (1 + torch.exp(-gb))).to(torch.bfloat16).double() * ub
        oracle.assert_close(act[None].cpu(), want[None].cpu().numpy(), "bf16", name)
    elif kind == "residual":
        r, wg = _plan(ops, m, _lib.GBX_DG_RESIDUAL)
        h = torch.randn(n).to(torch.bfloat16).to(DEV)
        h0 = h.clone()
        ssq = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=DEV) if wg > 0 else None
        ops.dgemv(ops.make_dgemv_desc(x, m, h, ops.dgemv_pack(m, r), r, residual=h, ssq_out=ssq), x.device)
        want = h0.double() + (deq(m) @ x.double())
        oracle.assert_close(h[None].cpu(), want[None].cpu().numpy(), "bf16", name)
        if ssq is not None:
            assert abs(ssq.double().sum().item() - h.double().pow(2).sum().item()) <= 1e-5 * h.double().pow(2).sum().item()
        if bits == 2 and name.startswith("8B"):
            assert ssq is not None, "the 2-bit o_proj / down_proj launches of the headline bench carry the statistic"
    else:
        r, wg = _plan(ops, m, _lib.GBX_DG_PLAIN, True)
        y = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        ops.dgemv(ops.make_dgemv_desc(x, m, y, ops.dgemv_pack(m, r), r, norm_weight=nw, eps=1e-5), x.device)
        want = deq(m) @ _rms(x, nw, 1e-5).double()
        oracle.assert_close(y[None].cpu(), want[None].cpu().numpy(), "bf16", name)


# ---- the matrix-core form of the same launches (csrc/dgemv_mx.hip; GBX_DG_MX): opt-in per process, so it runs in a child ------
_MX_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from gbx_lm_amd import ops, _lib
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
from oracle import gba_oracle as O            # checker only
DEV = "cuda:0"
def mk(n, k, seed):
    d = O.synth_layer(n, k, 2, 128, seed=seed)
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m, d
def rms(x, w, eps):
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()).to(x.dtype)
ref = lambda x, d: O.qlinear_ref(x[None], d["qweight"], d["scales"], d["zeros"], 2, 128, acc=np.float64)
torch.manual_seed(7)
for (n, k) in ((512, 4096), (14336, 4096), (1030, 2048), (640, 8192)):
    gate, dg = mk(n, k, 1); up, du = mk(n, k, 2)
    x = (torch.randn(k) * 2).to(torch.bfloat16).to(DEV)
    x[5] = 90.0                                                  # an outlier 2^6 above its block: the limbs must stay exact
    nw = (1 + 0.1 * torch.randn(k)).to(torch.bfloat16).to(DEV)
    xn = rms(x, nw, 1e-5).cpu()
    # SwiGLU + norm, statistic in-kernel and carried
    for carried in (False, True):
        ssq = None
        if carried:
            ssq = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=DEV); ssq[:256] = x.float().pow(2).view(256, -1).sum(1)
        act = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        ops.dgemv(ops.make_dgemv_desc(x, gate, act, ops.dgemv_pack(gate, 2, up), 2, mod_up=up, norm_weight=nw, eps=1e-5, ssq_in=ssq), x.device)
        g64, u64 = ref(xn, dg), ref(xn, du)
        O.assert_close(act[None], (g64 / (1 + np.exp(-g64))) * u64, "bf16", f"mx swiglu {n}x{k} carried={carried}")
    # plain + norm, residual with the partial statistic
    y = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    ops.dgemv(ops.make_dgemv_desc(x, gate, y, ops.dgemv_pack(gate, 2), 2, norm_weight=nw, eps=1e-5), x.device)
    O.assert_close(y[None], ref(xn, dg), "bf16", f"mx plain {n}x{k}")
    h = torch.randn(n).to(torch.bfloat16).to(DEV); h0 = h.clone()
    r, parts = ops.dgemv_plan(n, k, 2, 128, 1)
    ssq_o = torch.zeros(_lib.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=DEV)
    ok = ops.dgemv_plan(n, k, 2, 128, 1)[1] > 0
    ops.dgemv(ops.make_dgemv_desc(x, gate, h, ops.dgemv_pack(gate, 2), 2, residual=h), x.device)
    O.assert_close(h[None], h0.cpu().double().numpy()[None] + ref(x.cpu(), dg), "bf16", f"mx residual {n}x{k}")
print("MX_OK")
"""


def test_dgemv_matrix_core_form_vs_oracle():
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GBX_DG_MX="2")
    out = subprocess.run([sys.executable, "-c", _MX_CHILD % root], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and "MX_OK" in out.stdout, out.stderr[-3000:] + out.stdout[-500:]


# ---- edge cases of the matrix-core form (review of round 4: it is the dominant kernel of the token): index map, non-finite inputs,
#      the bound of the two-limb activation split, agreement with the vector form ----
_MX_EDGE_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from gbx_lm_amd import ops, _lib
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
DEV = "cuda:0"
def unit(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k // 16), dtype=torch.int64, generator=g).to(torch.int32).view(torch.uint32).to(DEV)
    m._buffers["scales"] = torch.ones((n, k // 128), dtype=torch.bfloat16, device=DEV)
    m._buffers["zeros"] = torch.zeros((n, k // 128), dtype=torch.bfloat16, device=DEV)
    return m
def run(m, x):
    y = torch.zeros(m.output_dims, dtype=torch.bfloat16, device=DEV)
    d = ops.make_dgemv_desc(x, m, y, ops.dgemv_pack(m, 2), 2)
    assert ops.dgemv_kernel_name(d).startswith("dgemv_mx_kernel"), ops.dgemv_kernel_name(d)
    ops.dgemv(d, x.device)
    return y
for k in (2048, 4096):                          # SP = 1, 2 spans: the shapes gbx_dgemv_forward puts on the matrix cores with the default
    n = 64                                      # switches (K = 8192 rows are two lane passes long and run in 8-wave workgroups, vector form)
    m = unit(n, k, k)
    q = ops.unpack(m.qweight, 2).to(torch.float32)             # [n, k] integer fields (bit-exact unpack, tested elsewhere)
    # (i) one-hot activations, unit scales: y[n] == q[n, k0] EXACTLY -- every position of the first span's first and last group,
    #     of the last span's last group, and a stride through the rest: pins the operand slot map of dgemv_mx.hip:12-13
    ks = list(range(0, 128)) + list(range(2048 - 128, 2048)) + list(range(k - 128, k)) + list(range(5, k, 97))
    for k0 in ks:
        for val in (1.0, -2.0, 0.00390625):
            x = torch.zeros(k, dtype=torch.bfloat16, device=DEV); x[k0] = val
            y = run(m, x)
            assert torch.equal(y.float(), (q[:, k0] * val).to(torch.bfloat16).float()), (k, k0, val)
    # (ii) all-zero x -> exact zeros; NaN / +-Inf anywhere -> every output non-finite (NaN: documented rule of the matrix-core form)
    assert int(run(m, torch.zeros(k, dtype=torch.bfloat16, device=DEV)).abs().sum()) == 0
    for bad in (float("nan"), float("inf"), float("-inf")):
        x = torch.randn(k).to(torch.bfloat16).to(DEV); x[k // 3] = bad
        assert not torch.isfinite(run(m, x).float()).any(), (k, bad)
    # (iii) the two-limb split: an element >= 2^-9 of its 32-value block's maximum is carried exactly, a smaller one with an absolute
    #       error <= 2^-17 of the block maximum (E4M3 limbs: 4 + 4 significant bits below a maximum scaled into [128, 256))
    for j in range(0, 21):
        x = torch.zeros(k, dtype=torch.bfloat16, device=DEV)
        x[64] = 1.0                                             # block maximum
        x[70] = 2.0 ** -j                                       # same block
        y = run(m, x).double()
        want = (q[:, 64] + q[:, 70] * 2.0 ** -j).double()
        err = (y - want.to(torch.bfloat16).double()).abs().max().item()
        lim = 0.0 if j <= 9 else 3 * 2.0 ** -17
        ulp = 2.0 ** -7 * max(1.0, want.abs().max().item())     # one bf16 rounding of the output
        assert err <= lim + ulp * (j > 9), (k, j, err)
        if j <= 9:
            assert torch.equal(y.float(), want.to(torch.bfloat16).float()), (k, j)
    # wide in-block dynamic range, random signs: against fp64 with the bound 3 * 32 * 2^-17 * bmax per block + output rounding
    g = torch.Generator().manual_seed(k + 1)
    x = (torch.randn(k, generator=g) * torch.exp2(-torch.randint(0, 21, (k,), generator=g).float())).to(torch.bfloat16)
    bmax = x.float().abs().view(-1, 32).max(1).values
    y = run(m, x.to(DEV)).double().cpu()
    want = q.double().cpu() @ x.double()
    bound = 3 * 32 * 2.0 ** -17 * bmax.double().sum().item()
    assert ((y - want).abs() <= bound + 2.0 ** -7 * want.abs() + 1e-6).all(), (k, (y - want).abs().max().item(), bound)
# (iv) agreement with the vector form on ordinary inputs (random scales / zeros): <= 1 bf16 ulp of the larger of the two everywhere
from oracle import gba_oracle as O
for (n, k) in ((256, 4096), (128, 2048)):
    d = O.synth_layer(n, k, 2, 128, seed=3)
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    x = torch.randn(k).to(torch.bfloat16).to(DEV)
    y_mx = run(m, x).float()
    y_v = ops.quantized_matmul(x[None], m.qweight, m.scales, m.zeros, True, 128, 2)[0].float()      # gemv_kernel (vector dot)
    ulp = torch.maximum(y_mx.abs(), y_v.abs()) * 2.0 ** -7
    rms = y_v.pow(2).mean().sqrt().item()
    # both round an fp32-accurate sum: one rounding step apart where the sum sits at a rounding boundary, plus -- on sums that cancel to
    # near zero -- the two forms' own accumulation error (2^-10 of the vector's RMS covers it; the oracle tolerance is 2e-2 of the RMS)
    assert ((y_mx - y_v).abs() <= ulp + 2.0 ** -10 * rms).all(), (n, k, (y_mx - y_v).abs().max().item(), rms)
    assert (y_mx == y_v).float().mean().item() > 0.8, (n, k, (y_mx == y_v).float().mean().item())
    want = O.qlinear_ref(x.cpu()[None], d["qweight"], d["scales"], d["zeros"], 2, 128, acc=np.float64)[0]
    tight = np.abs(y_mx.cpu().double().numpy() - want) <= 2.0 ** -7 * np.abs(want) + 2.0 ** -7 * np.sqrt((want ** 2).mean()) * 0.05
    assert tight.mean() > 0.99, (n, k, tight.mean())
print("MX_EDGE_OK")
"""


def test_dgemv_matrix_core_form_edge_cases():
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GBX_DG_MX="2")
    out = subprocess.run([sys.executable, "-c", _MX_EDGE_CHILD % root], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and "MX_EDGE_OK" in out.stdout, out.stderr[-3000:] + out.stdout[-500:]
