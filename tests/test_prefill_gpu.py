"""GPU: BASELINE.json configs[3] (prefill seq 4096 x batch 8 -> M = 32768 rows) through gbx_qlinear_forward, on the
matrix-core dequant-GEMM tiles that M actually dispatches to (csrc/mfma_gemm.hip: 256 x 256 with 8 waves, 256 x 128,
128 x 128, and 128 x 128 with k split over workgroups).  Every case asserts the dispatched tile
(gbx_qlinear_kernel_id) and compares
  (a) a strided subset of rows against the C oracle with fp64 accumulation (oracle.tolerance), and
  (b) ALL rows against x . W^T in fp64 with W from gbx_dequantize (bit-exact vs the oracle, test_qlinear_gpu.py /
      test_fullsize_gpu.py::test_dequantize_full_size_rows_vs_c_oracle).
Reference op: mx.quantized_matmul at gbx_lm/models/quantized_linear_gba.py:195-205, called with M = B * L rows by the
prefill of gbx_lm/utils.py:312-319."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TILE = {11: "128x128", 12: "256x128", 13: "256x256-8w", 15: "128x128 split-K", 31: "dense 128x128", 32: "dense 256x128",
        33: "dense 256x256-8w"}


def _layer(n, k, bits, g, seed, dtype=torch.bfloat16):
    gen = torch.Generator(device=DEV)
    gen.manual_seed(seed)
    qw = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=DEV, generator=gen).to(torch.int32).view(torch.uint32)
    qmax = float((1 << bits) - 1)
    sc = ((torch.rand((n, k // g), device=DEV, generator=gen) + 0.5) * (2.0 / qmax) * 0.02)
    ze = -sc * qmax / 2 + torch.randn((n, k // g), device=DEV, generator=gen) * 0.1 * sc
    return qw, sc.to(dtype), ze.to(dtype)


def _check(oracle, c_oracle, m, n, k, bits, g, dtype, want_tile, bias=False, workspace=False, seed=0):
    from gbx_lm_amd import _lib, ops
    name = "bf16" if dtype == torch.bfloat16 else "fp16"
    kid = _lib.lib().gbx_qlinear_kernel_id(m, n, k, bits, g, int(workspace))
    assert kid == want_tile, f"M{m} N{n} K{k}: dispatched to {TILE.get(kid, kid)}, test expects {TILE[want_tile]}"
    qw, sc, ze = _layer(n, k, bits, g, seed + m + n + bits, dtype)
    gen = torch.Generator(device=DEV)
    gen.manual_seed(seed + 1)
    x = torch.randn((m, k), device=DEV, generator=gen).to(dtype)
    b = (torch.randn((n,), device=DEV, generator=gen) * 0.5).to(dtype) if bias else None
    L, dt = _lib.lib(), (_lib.GBX_BF16 if dtype == torch.bfloat16 else _lib.GBX_F16)
    y = torch.empty((m, n), dtype=dtype, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    if workspace:
        need = L.gbx_qlinear_workspace_bytes(m, n, k, bits, g)
        assert need > 0
        ws = torch.empty((need,), dtype=torch.uint8, device=DEV)
        _lib.check(L.gbx_qlinear_forward_ws(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), b.data_ptr() if bias else None,
                                            y.data_ptr(), m, n, k, bits, g, dt, ws.data_ptr(), need, stream))
    else:                                                         # straight through the C-ABI, no scratch: one launch
        _lib.check(L.gbx_qlinear_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), b.data_ptr() if bias else None,
                                         y.data_ptr(), m, n, k, bits, g, dt, stream))
    torch.cuda.synchronize()
    assert y.shape == (m, n) and torch.isfinite(y.float()).all()
    # (b) every row vs fp64 x . dequantised weights (+ bias, second rounding like the reference's separate add)
    w = ops.dequantize(qw, sc, ze, group_size=g, bits=bits, dtype=torch.float32).double()
    rtol, _ = oracle.tolerance(np.ones(1), name)
    worst = 0.0
    for r0 in range(0, m, 8192):                                  # fp64 reference in slabs (memory)
        ref = x[r0:r0 + 8192].double() @ w.t()
        if bias:
            ref = ref + b.double()[None]
        rms = ref.pow(2).mean().sqrt().item()
        err = (y[r0:r0 + 8192].double() - ref).abs()
        bound = rtol * rms + rtol * ref.abs()
        assert (err <= bound).all(), (f"M{m} N{n} K{k} b{bits} g{g} {name} rows {r0}..: {(err > bound).sum().item()} outside "
                                      f"tolerance, max err {err.max().item():.4g}, rms {rms:.4g}")
        worst = max(worst, (err / bound).max().item())
    # (a) strided row subset vs the C oracle (fp64 accumulation), the stated tolerance
    rows = sorted({0, 1, m - 1, m // 2, *range(5, m, max(m // 11, 1))})[:16]
    xs = x[rows].cpu()
    y64 = c_oracle.qlinear(xs, qw.cpu().view(torch.int32).numpy().view(np.uint32), sc.cpu(), ze.cpu(), bits, g,
                           bias=b.cpu() if bias else None, acc64=True)
    oracle.assert_close(y[rows], y64, name, f"prefill M{m} N{n} K{k} b{bits} g{g} vs C oracle")
    return worst


# the four QuantizedLinear shapes of an 8B block at M = 32768 (4-bit = S40, the cfg4 strategy): 256 x 256 / 8 waves everywhere
@pytest.mark.parametrize("n,k", [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)])
def test_cfg4_full_m_4bit(oracle, c_oracle, n, k):
    _check(oracle, c_oracle, 32768, n, k, 4, 128, torch.bfloat16, want_tile=13)


@pytest.mark.parametrize("bits", [2, 3])
def test_cfg4_full_m_other_widths(oracle, c_oracle, bits):
    _check(oracle, c_oracle, 32768, 4096, 4096, bits, 128, torch.bfloat16, want_tile=13)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tile_256x256_8w(oracle, c_oracle, bits, dtype):
    g = 64 if dtype == torch.float16 else 128
    _check(oracle, c_oracle, 4096, 4096, 4096, bits, g, dtype, want_tile=13, bias=(bits == 3))
    _check(oracle, c_oracle, 4096 - 37, 14336 - 24, 4096, bits, g, dtype, want_tile=13)          # ragged M and N


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tile_256x128(oracle, c_oracle, bits, dtype):
    g = 128 if dtype == torch.float16 else 64
    _check(oracle, c_oracle, 4096, 1024, 4096, bits, g, dtype, want_tile=12)
    _check(oracle, c_oracle, 1024, 4096, 14336, bits, g, dtype, want_tile=12, bias=(bits == 2))
    _check(oracle, c_oracle, 1000, 14336 - 8, 4096, bits, g, dtype, want_tile=12)               # ragged


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_tile_128x128_and_splitk(oracle, c_oracle, bits):
    _check(oracle, c_oracle, 1024, 1024, 4096, bits, 128, torch.bfloat16, want_tile=11)
    _check(oracle, c_oracle, 100, 4096 - 12, 4096, bits, 64, torch.float16, want_tile=11, bias=True)
    _check(oracle, c_oracle, 1000, 1024, 4096, bits, 128, torch.bfloat16, want_tile=15, workspace=True)
    _check(oracle, c_oracle, 300, 4096, 14336, bits, 128, torch.bfloat16, want_tile=15, workspace=True, bias=True)


# M >= 4096 with caller scratch: the weights are dequantised ONCE per call into the workspace and the same tiles run on them
# (what gbx_lm_amd.ops.quantized_matmul does for every long prompt)
@pytest.mark.parametrize("n,k", [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)])
def test_cfg4_full_m_4bit_dense(oracle, c_oracle, n, k):
    _check(oracle, c_oracle, 32768, n, k, 4, 128, torch.bfloat16, want_tile=33, workspace=True)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dense_tiles(oracle, c_oracle, bits, dtype):
    g = 64 if dtype == torch.float16 else 128
    _check(oracle, c_oracle, 8192, 4096, 4096, bits, g, dtype, want_tile=33, workspace=True, bias=(bits == 3))
    _check(oracle, c_oracle, 8192 + 19, 14336 - 24, 4096, bits, g, dtype, want_tile=33, workspace=True)      # ragged M and N
    _check(oracle, c_oracle, 4096, 1024, 4096, bits, g, dtype, want_tile=32, workspace=True)


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_dense_ksplit_mid_prompts(oracle, c_oracle, bits):
    """1024..4095-row chunks whose 256 x 256 tiles leave most CUs idle: the dense GEMM with its k range split over workgroups
    (fp32 slices in the workspace tail + the summing launch), through gbx_qlinear_forward_ws"""
    from gbx_lm_amd import _lib
    L = _lib.lib()
    assert L.gbx_dense_workspace_bytes(2048, 4096, 4096) == 2 * 2048 * 4096 * 4 and L.gbx_dense_workspace_bytes(8192, 4096, 4096) == 0
    _check(oracle, c_oracle, 2048, 4096, 4096, bits, 128, torch.bfloat16, want_tile=32, workspace=True)
    _check(oracle, c_oracle, 1024, 4096, 14336, bits, 64, torch.float16, want_tile=32, workspace=True, bias=True)
    _check(oracle, c_oracle, 1500, 4096 - 12, 4096, bits, 128, torch.bfloat16, want_tile=32, workspace=True)      # ragged M and N
    _check(oracle, c_oracle, 1024 + 77, 1024, 4096, bits, 128, torch.bfloat16, want_tile=31, workspace=True)


def test_dense_path_close_to_fused_tiles(oracle):
    """same weights rounded the same way (one gbx_dequantize rounding either way); the dense kernel accumulates in another
    order (16x16x32 tiles), so equality holds up to the fp32 accumulation -> output-dtype rounding: a few ulps at most"""
    from gbx_lm_amd import _lib
    L = _lib.lib()
    m, n, k, bits, g = 8192, 4096, 4096, 4, 128
    qw, sc, ze = _layer(n, k, bits, g, 5)
    x = torch.randn((m, k), device=DEV).to(torch.bfloat16)
    y0, y1 = torch.empty((m, n), dtype=torch.bfloat16, device=DEV), torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.gbx_qlinear_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, y0.data_ptr(), m, n, k, bits, g, _lib.GBX_BF16, st))
    need = L.gbx_qlinear_workspace_bytes(m, n, k, bits, g)
    assert need == n * k * 2
    ws = torch.empty((need,), dtype=torch.uint8, device=DEV)
    _lib.check(L.gbx_qlinear_forward_ws(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, y1.data_ptr(), m, n, k, bits, g, _lib.GBX_BF16,
                                        ws.data_ptr(), need, st))
    torch.cuda.synchronize()
    d = (y0.float() - y1.float()).abs()
    rms = y0.float().pow(2).mean().sqrt().item()
    assert (d <= 2.0 ** -7 * y0.float().abs() + 1e-3 * rms).all(), f"max diff {d.max().item()} (rms {rms})"
    # repeated launches are deterministic (no race between the DMA stages and the fragment reads shows up as a diff here)
    for _ in range(5):
        y2 = torch.empty_like(y1)
        _lib.check(L.gbx_qlinear_forward_ws(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, y2.data_ptr(), m, n, k, bits, g,
                                            _lib.GBX_BF16, ws.data_ptr(), need, st))
        torch.cuda.synchronize()
        assert torch.equal(y1, y2)


@pytest.mark.parametrize("bits,dtype,m,n", [(4, torch.bfloat16, 16384, 14336), (2, torch.float16, 4096 + 77, 1024), (3, torch.bfloat16, 4096, 256)])
def test_dense_swiglu_epilogue(bits, dtype, m, n):
    """gbx_qlinear_swiglu_forward_ws at M >= 4096 (gate | up dequantised into one dense matrix, SwiGLU in the GEMM epilogue)
    == gbx_swiglu(gate GEMM, up GEMM) through the same dense path, bit for bit: every output element accumulates over k in the
    same order whichever tile column it sits in, and the epilogue applies the same roundings to the rounded GEMM outputs"""
    from gbx_lm_amd import _lib, ops
    L = _lib.lib()
    k, g = 4096, 128
    dt = _lib.GBX_BF16 if dtype == torch.bfloat16 else _lib.GBX_F16
    gq, gs, gz = _layer(n, k, bits, g, 11, dtype)
    uq, us, uz = _layer(n, k, bits, g, 12, dtype)
    x = torch.randn((m, k), device=DEV).to(dtype)
    st = torch.cuda.current_stream().cuda_stream
    need = L.gbx_qlinear_swiglu_workspace_bytes(m, n, k)
    assert need == 2 * n * k * 2
    ws = torch.empty((need,), dtype=torch.uint8, device=DEV)
    y = torch.empty((m, n), dtype=dtype, device=DEV)
    _lib.check(L.gbx_qlinear_swiglu_forward_ws(x.data_ptr(), gq.data_ptr(), gs.data_ptr(), gz.data_ptr(), uq.data_ptr(), us.data_ptr(),
                                               uz.data_ptr(), y.data_ptr(), m, n, k, bits, g, dt, ws.data_ptr(), need, st))
    outs = []
    for (q, s, z) in ((gq, gs, gz), (uq, us, uz)):
        o = torch.empty((m, n), dtype=dtype, device=DEV)
        nd = L.gbx_qlinear_workspace_bytes(m, n, k, bits, g)
        w1 = torch.empty((nd,), dtype=torch.uint8, device=DEV)
        _lib.check(L.gbx_qlinear_forward_ws(x.data_ptr(), q.data_ptr(), s.data_ptr(), z.data_ptr(), None, o.data_ptr(), m, n, k, bits, g, dt,
                                            w1.data_ptr(), nd, st))
        outs.append(o)
    ref = ops.swiglu(outs[0], outs[1])
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    if L.gbx_dense_workspace_bytes(m, n, k) == 0:
        assert torch.equal(y, ref), f"{(y != ref).sum().item()} of {y.numel()} elements differ, max {(y.float() - ref.float()).abs().max().item()}"
    else:       # the single GEMMs of this few-tile shape split k over workgroups (another summation order): equal up to output ulps
        d = (y.float() - ref.float()).abs()
        rms = ref.float().pow(2).mean().sqrt().item()
        assert (d <= 2.0 ** -6 * ref.float().abs() + 2e-3 * rms).all() and (d > 0).float().mean().item() < 0.1


def test_mlp_long_prompt_uses_the_fused_path(monkeypatch):
    from gbx_lm_amd import ops, synthetic as S
    cfg = dict(S.TINY, hidden_size=1024, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2)
    model = S.build_model(cfg, S.make_strategy("S40", cfg["num_hidden_layers"], 128), device=torch.device(DEV), seed=2)
    mlp = model.model.layers[0].mlp
    x = torch.randn((2, 2048, 1024), device=DEV).to(torch.bfloat16)
    a = mlp(x)
    monkeypatch.setattr(ops, "quantized_swiglu", lambda *a, **k: None)
    b = mlp(x)
    # (the separate GEMMs of this small shape split k over workgroups: another summation order, so a few activations differ by an
    # output ulp and the down projection -- a 1024-term sum with cancellation -- spreads that over its outputs)
    d = (a.float() - b.float()).abs()
    assert (d <= 2.0 ** -5 * b.float().abs() + 3e-2 * b.float().pow(2).mean().sqrt()).all()


def test_resident_dense_images():
    """ops.quantized_matmul / quantized_swiglu at M >= 4096 keep the dequantised image of a call and reuse it for the next call
    on the SAME tensor objects (bit-identical output); a modified or re-created tensor never sees a stale image"""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    ops.dense_cache_clear()
    m, n, k, bits, g = 4096, 1024, 4096, 4, 128
    qw, sc, ze = _layer(n, k, bits, g, 21)
    x = torch.randn((m, k), device=DEV).to(torch.bfloat16)
    y0 = ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=bits)
    assert len(ops._dense_cache.entries) == 1 and ops._dense_cache.nbytes == n * k * 2
    assert ops._dense_cache.get((qw, sc, ze), (n, k, bits, g, x.dtype, 0)) is not None
    y1 = ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=bits)            # served from the resident image
    assert torch.equal(y0, y1)
    sc.mul_(2.0)                                                                  # in place: the version counter invalidates the image
    y2 = ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=bits)
    w2 = ops.dequantize(qw, sc, ze, group_size=g, bits=bits, dtype=torch.float32)
    ref = x.float() @ w2.t()
    assert (y2.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item()
    assert not torch.equal(y2, y0)
    qw2, sc2, ze2 = qw.clone(), (sc * 0.5).clone(), ze.clone()                     # other objects: a miss, not a stale hit
    y3 = ops.quantized_matmul(x, qw2, sc2, ze2, group_size=g, bits=bits)
    assert torch.equal(y3, ops.quantized_matmul(x, qw, sc2, ze, group_size=g, bits=bits))
    # gate | up image
    ops.dense_cache_clear()
    gate, up = QuantizedLinear(k, n, bias=False, group_size=g, bits=bits), QuantizedLinear(k, n, bias=False, group_size=g, bits=bits)
    for mod, seed in ((gate, 31), (up, 32)):
        q_, s_, z_ = _layer(n, k, bits, g, seed)
        mod._set("qweight", q_), mod._set("scales", s_), mod._set("zeros", z_)
    a0 = ops.quantized_swiglu(x, gate, up)
    assert len(ops._dense_cache.entries) == 1
    a1 = ops.quantized_swiglu(x, gate, up)
    assert torch.equal(a0, a1)
    want = ops.swiglu(ops.quantized_matmul(x, gate.qweight, gate.scales, gate.zeros, group_size=g, bits=bits),
                      ops.quantized_matmul(x, up.qweight, up.scales, up.zeros, group_size=g, bits=bits))
    d = (a0.float() - want.float()).abs()           # (the single GEMMs of this few-tile shape split k: equal up to output ulps)
    assert (d <= 2.0 ** -6 * want.float().abs() + 2e-3 * want.float().pow(2).mean().sqrt()).all()
    old = ops._dense_cache.budget
    try:                                                                          # a budget of 0 disables it
        ops._dense_cache.budget = 0
        ops.dense_cache_clear()
        ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=bits)
        assert len(ops._dense_cache.entries) == 0
    finally:
        ops._dense_cache.budget = old
        ops.dense_cache_clear()
