"""GPU parity tests proper: the HIP path through the C-ABI vs the oracle on the same seeded inputs.
Bit-exact for the integer unpack / indexing; stated fp tolerance (oracle.tolerance) for the matmul."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _np_u32(t):
    return t.cpu().view(torch.int32).numpy().view(np.uint32)


def _dev(d):
    return {k: (torch.from_numpy(v.view(np.int32)).view(torch.uint32).to(DEV) if isinstance(v, np.ndarray)
                else (v.to(DEV) if v is not None else None)) for k, v in d.items()}


def _h(a, dt):
    return torch.from_numpy(a.view(np.int16)).view(torch.bfloat16 if dt == "bf16" else torch.float16)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from gbx_lm_amd import ops as o, _lib
    assert os.path.exists(_lib.so_path()), "HIP extension must be built in-tree"
    _lib.lib()
    return o


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_unpack_bit_exact(ops, oracle, bits):
    rng = np.random.default_rng(bits)
    for (n, k) in [(3, 32), (5, 96), (17, 4096), (2, 14336)]:
        qw = rng.integers(0, 2**32, size=(n, k * bits // 32), dtype=np.uint64).astype(np.uint32)
        got = ops.unpack(torch.from_numpy(qw.view(np.int32)).view(torch.uint32).to(DEV), bits).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.unpack(qw, bits))


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("group", [32, 64, 128])
def test_dequantize_fp32_bit_exact(ops, oracle, bits, group):
    d = oracle.synth_layer(24, 512, bits, group, seed=bits * 7 + group)
    dd = _dev(d)
    w = ops.dequantize(dd["qweight"], dd["scales"], dd["zeros"], group_size=group, bits=bits, dtype=torch.float32)
    np.testing.assert_array_equal(w.cpu().numpy(), oracle.dequant(d["qweight"], d["scales"], d["zeros"], bits, group))
    wb = ops.dequantize(dd["qweight"], dd["scales"], dd["zeros"], group_size=group, bits=bits)
    want = torch.from_numpy(oracle.dequant(d["qweight"], d["scales"], d["zeros"], bits, group)).to(torch.bfloat16)
    assert torch.equal(wb.cpu(), want)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("k", [128, 4096, 14336])
def test_gemv_integer_indexing_exact(ops, oracle, bits, k):
    """One-hot activations, unit scales, zero zeros: y[n] == q[n, k0] exactly, so every in-register extraction
    path of the streaming GEMV (all 128 positions of a period, first / middle / last slot) is checked bit-exactly."""
    rng = np.random.default_rng(100 + bits)
    n, g = 16, 128
    qw = rng.integers(0, 2**32, size=(n, k * bits // 32), dtype=np.uint64).astype(np.uint32)
    q = oracle.unpack(qw, bits)
    qw_d = torch.from_numpy(qw.view(np.int32)).view(torch.uint32).to(DEV)
    sc = torch.ones((n, k // g), dtype=torch.bfloat16, device=DEV)
    ze = torch.zeros((n, k // g), dtype=torch.bfloat16, device=DEV)
    positions = sorted(set(list(range(0, 128)) + list(range(k - 128, k)) + list(range(k // 2, min(k, k // 2 + 64)))))
    x = torch.zeros((len(positions), k), dtype=torch.bfloat16, device=DEV)
    for i, p in enumerate(positions):
        x[i, p] = 1.0
    for i, p in enumerate(positions):      # M = 1 calls: the decode GEMV itself
        y = ops.quantized_matmul(x[i:i + 1], qw_d, sc, ze, group_size=g, bits=bits)
        np.testing.assert_array_equal(y.float().cpu().numpy()[0], q[:, p].astype(np.float32), err_msg=f"k={p}")


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("group", [32, 64, 128])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_golden_small_cases(ops, oracle, golden_dir, bits, group, dt):
    """Committed fixtures (K = 256: generic kernel) through the C-ABI."""
    z = np.load(os.path.join(golden_dir, "qlinear_small.npz"))
    key = f"b{bits}_g{group}_{dt}"
    qw = torch.from_numpy(z[key + "_qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    sc, ze, x = (_h(z[key + s], dt).to(DEV) for s in ("_scales", "_zeros", "_x"))
    bias = _h(z[key + "_bias"], dt).to(DEV) if key + "_bias" in z.files else None
    y = ops.quantized_matmul(x, qw, sc, ze, group_size=group, bits=bits, bias=bias)
    oracle.assert_close(y, z[key + "_y64"], dt, key)
    for i in range(x.shape[0]):
        yi = ops.quantized_matmul(x[i:i + 1], qw, sc, ze, group_size=group, bits=bits, bias=bias)
        oracle.assert_close(yi, z[key + "_y64"][i:i + 1], dt, key + f"[{i}]")


SHAPES = [  # (N, K): config shapes of Llama-3-8B / 3.2-3B (rows trimmed where the oracle would take long) + odd ones
    (4096, 4096), (1024, 4096), (512, 14336), (1536, 3072), (256, 8192), (1000, 1792), (37, 512), (64, 128), (5, 2048),
]


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("n,k", SHAPES)
def test_gemv_matches_oracle(ops, oracle, c_oracle, bits, n, k):
    for group in (128, 32) if k % 128 == 0 else (64,):
        d = oracle.synth_layer(n, k, bits, group, seed=n + k + bits + group, m=1, bias=(n % 2 == 1))
        dd = _dev(d)
        y = ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=group, bits=bits, bias=dd["bias"])
        y64 = c_oracle.qlinear(d["x"], d["qweight"], d["scales"], d["zeros"], bits, group, bias=d["bias"], acc64=True)
        oracle.assert_close(y, y64, "bf16", f"N{n} K{k} b{bits} g{group}")
        # tighter: before the single output rounding the kernel is an fp32 computation; its rounded result must be
        # within one bf16 ulp-ish of the rounded fp64 result almost everywhere
        ref = torch.from_numpy(y64.astype(np.float32)).to(torch.bfloat16).float().numpy()
        got = y.float().cpu().numpy()
        rms = np.sqrt(np.mean(y64 ** 2))
        assert np.abs(got - ref).max() <= 2 ** -7 * max(np.abs(y64).max(), rms) + 1e-6


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_f16_and_extreme_activation_ranges(ops, oracle, c_oracle, bits):
    """Block-exponent path: tiny, huge and mixed-magnitude activations (bf16), and fp16 activations."""
    n, k, g = 64, 4096, 128
    for scale, dt in [(1e-20, "bf16"), (1e20, "bf16"), (1.0, "f16"), (300.0, "f16"), (1e-3, "f16")]:
        d = oracle.synth_layer(n, k, bits, g, seed=bits, act_dtype=dt)
        x = d["x"].float() * scale
        x[0, ::7] *= 1e-4           # wide dynamic range inside one vector
        d["x"] = x.to(d["x"].dtype)
        dd = _dev(d)
        y = ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=g, bits=bits)
        y64 = c_oracle.qlinear(d["x"], d["qweight"], d["scales"], d["zeros"], bits, g, acc64=True)
        oracle.assert_close(y, y64, dt, f"scale={scale} {dt}")
    # all-zero activations -> zeros exactly; NaN propagates
    dd["x"].zero_()
    y = ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=g, bits=bits)
    assert torch.count_nonzero(y) == 0
    dd["x"][0, 5] = float("nan")
    y = ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=g, bits=bits)
    assert torch.isnan(y).all()


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("m", [2, 5, 8, 33])
def test_batched_rows_match_oracle(ops, oracle, c_oracle, bits, m):
    n, k, g = 192, 1024, 64
    d = oracle.synth_layer(n, k, bits, g, seed=m * 10 + bits, m=m, bias=True)
    dd = _dev(d)
    y = ops.quantized_matmul(dd["x"].reshape(1, m, k), dd["qweight"], dd["scales"], dd["zeros"], group_size=g, bits=bits, bias=dd["bias"])
    assert y.shape == (1, m, n)
    y64 = c_oracle.qlinear(d["x"], d["qweight"], d["scales"], d["zeros"], bits, g, bias=d["bias"], acc64=True)
    oracle.assert_close(y.reshape(m, n), y64, "bf16", f"M={m}")


def test_empty_batch_and_errors(ops, oracle):
    d = oracle.synth_layer(8, 256, 4, 64, seed=1)
    dd = _dev(d)
    y = ops.quantized_matmul(dd["x"][:0], dd["qweight"], dd["scales"], dd["zeros"], group_size=64, bits=4)
    assert y.shape == (0, 8)
    with pytest.raises(ValueError):      # wrong bits for this packing -> K mismatch
        ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=64, bits=2)
    with pytest.raises(ValueError, match="must be 32, 64 or 128"):
        ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"][:, :1].repeat(1, 16), dd["zeros"][:, :1].repeat(1, 16), group_size=16, bits=4)
    with pytest.raises(ValueError):
        ops.quantized_matmul(dd["x"].float(), dd["qweight"], dd["scales"], dd["zeros"], group_size=64, bits=4)
    with pytest.raises(NotImplementedError):
        ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], transpose=False, group_size=64, bits=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.quantized_matmul(d["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=64, bits=4)


def test_module_forward_and_graph_capture(ops, oracle, c_oracle):
    """QuantizedLinear module (reference signature) + the call is HIP-graph capturable (no sync/alloc inside)."""
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    n, k, b, g = 256, 4096, 2, 128
    d = oracle.synth_layer(n, k, b, g, seed=3)
    m = QuantizedLinear(k, n, bias=False, group_size=g, bits=b).to(DEV)
    dd = _dev(d)
    m._buffers["qweight"], m._buffers["scales"], m._buffers["zeros"] = dd["qweight"], dd["scales"], dd["zeros"]
    y64 = c_oracle.qlinear(d["x"], d["qweight"], d["scales"], d["zeros"], b, g, acc64=True)
    y = m(dd["x"])
    oracle.assert_close(y, y64, "bf16", "module")
    out = torch.empty((1, n), dtype=torch.bfloat16, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=g, bits=b, out=out)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(gr):
        ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=g, bits=b, out=out)
    out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, y)


@pytest.mark.parametrize("bits,g", [(2, 128), (3, 32), (4, 64)])
@pytest.mark.parametrize("m,n,k", [(20, 4128, 1024), (40, 4128, 512), (16, 200, 2048), (64, 96, 4096), (3, 4104, 4096)])
def test_skinny_mfma_tiles(ops, oracle, c_oracle, bits, g, m, n, k):
    """2 <= M <= 64: split-K MFMA kernel -- 32-row tiles (N > 4096) and 16-row tiles, 4- and 8-way K split, wide and
    per-value scale loads, ragged N and M."""
    d = oracle.synth_layer(n, k, bits, g, seed=m + bits, m=m)
    dd = _dev(d)
    y = ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=g, bits=bits)
    y64 = c_oracle.qlinear(d["x"], d["qweight"], d["scales"], d["zeros"], bits, g, acc64=True)
    oracle.assert_close(y, y64, "bf16", f"skinny M={m} N={n} K={k} b{bits} g{g}")


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("m,n,k", [(32, 24576, 1024), (7, 12288, 2048), (16, 6144, 3072), (2, 4096, 4096), (32, 1000, 1024),
                                   (25, 4096, 14336), (33, 24576, 1024), (64, 28672, 1024)])
def test_batch_kernel_no_dequant(ops, oracle, c_oracle, bits, dt, m, n, k):
    """2 <= M <= 32, group 128: integer fields straight to the matrix cores, affine map per group on the accumulators
    (mfma_batch.hip) -- every workgroup shape (128 / 64 / 32 / 16 rows), 1..14 steps, ragged N and M, both activation
    dtypes, bias as a second rounding."""
    d = oracle.synth_layer(n, k, bits, 128, seed=m + bits + n % 13, m=m, act_dtype=dt, bias=(n == 6144))
    x = d["x"].clone()
    x[0, 5] = 90.0                       # an outlier: the per-group offset term (C * xsum) must cancel exactly enough
    d["x"] = x
    dd = _dev(d)
    y = ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=128, bits=bits, bias=dd.get("bias"))
    y64 = c_oracle.qlinear(d["x"], d["qweight"], d["scales"], d["zeros"], bits, 128, acc64=True, bias=d.get("bias"))
    oracle.assert_close(y, y64, dt, f"batch M={m} N={n} K={k} b{bits} {dt}")


@pytest.mark.parametrize("g", [64, 32])
@pytest.mark.parametrize("m,n,k", [(32, 24576, 1024), (9, 12288, 2048), (16, 6144, 3072), (5, 4096, 4096), (32, 1000, 1024),
                                   (20, 4096, 14336)])
def test_batch_kernel_small_groups(ops, oracle, c_oracle, g, m, n, k):
    """2-bit weights with groups of 64 / 32 on the no-dequant kernel: the k assignment is interleaved so that every MFMA
    step stays inside one group, 2 / 4 affine flushes per chunk; every workgroup shape, ragged N."""
    for dt in ("bf16", "f16"):
        d = oracle.synth_layer(n, k, 2, g, seed=m + g + n % 11, m=m, act_dtype=dt)
        x = d["x"].clone()
        x[0, 37] = -70.0
        d["x"] = x
        dd = _dev(d)
        y = ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=g, bits=2)
        y64 = c_oracle.qlinear(d["x"], d["qweight"], d["scales"], d["zeros"], 2, g, acc64=True)
        oracle.assert_close(y, y64, dt, f"batch g{g} M={m} N={n} K={k} {dt}")


@pytest.mark.experimental
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("n,k", [(4096, 4096), (130, 1024), (1000, 2048), (4096, 14336), (28672, 4096)])
def test_tiled_matrix_core_gemv(ops, oracle, c_oracle, dt, n, k):
    """Decode GEMV on the int8 matrix cores from the load-time tiled form: same contract as the M = 1 forward
    (vs the fp64-accumulated oracle), ragged N, long rows, outlier activations (block fixed point must stay exact)."""
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    d = oracle.synth_layer(n, k, 2, 128, seed=n % 97 + k % 89, act_dtype=dt)
    x = d["x"].clone()
    x[0, 7] = 200.0                      # an outlier 2^7.6 above the rest of its group
    x[0, 300] = 1e-4
    d["x"] = x
    dd = _dev(d)
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"], m._buffers["scales"], m._buffers["zeros"] = dd["qweight"], dd["scales"], dd["zeros"]
    wt, szt = ops.prepack_tiles(m)
    y = ops.gemv_tiled(dd["x"].reshape(-1), wt, szt, n, k)
    y64 = c_oracle.qlinear(d["x"], d["qweight"], d["scales"], d["zeros"], 2, 128, acc64=True)
    oracle.assert_close(y[None], y64, dt, f"tiled gemv N={n} K={k}")
    yref = ops.quantized_matmul(dd["x"], dd["qweight"], dd["scales"], dd["zeros"], group_size=128, bits=2)
    assert (y[None].float() - yref.float()).abs().max().item() <= 2e-2 * yref.float().abs().max().item()
    with pytest.raises(NotImplementedError):
        m4 = QuantizedLinear(k, 16, False, 128, 4)
        m4._buffers["qweight"] = torch.zeros((16, k // 8), dtype=torch.int32, device=DEV).view(torch.uint32)
        m4._buffers["scales"] = torch.zeros((16, k // 128), dtype=dd["scales"].dtype, device=DEV)
        m4._buffers["zeros"] = torch.zeros((16, k // 128), dtype=dd["scales"].dtype, device=DEV)
        ops.prepack_tiles(m4)


@pytest.mark.parametrize("bits,g", [(2, 128), (3, 64), (4, 32)])
@pytest.mark.parametrize("m", [2, 32, 50])
def test_multi_matrix_launch_equals_separate_calls(ops, oracle, bits, g, m):
    """gbx_qlinear_forward_multi (q/k/v or gate/up in one split-K launch) is bit-identical to one call per matrix."""
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    k = 1024
    mods = []
    for i, n in enumerate((2048, 512, 4128 if m == 2 else 96)):       # total N > 4096 (32-row tiles) or <= 4096 (16-row)
        d = oracle.synth_layer(n, k, bits, g, seed=10 * i + bits, bias=(i == 1))
        q = QuantizedLinear(k, n, i == 1, g, bits)
        q._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
        q._buffers["scales"], q._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
        if i == 1:
            q._buffers["bias"] = d["bias"].to(DEV)
        mods.append(q)
    x = torch.randn(m, k).to(torch.bfloat16).to(DEV)
    fused = ops.quantized_matmul_multi(x, mods)
    # the no-dequant kernel (2/4 bits, g128, M <= 32) picks its k split from the total row count: same sums, another
    # fp32 association -> equal to an ulp of the output dtype instead of bit-identical
    def same(u, v):
        if g == 128 and m <= 64:
            return torch.allclose(u.float(), v.float(), rtol=2 ** -7, atol=2e-3)
        return torch.equal(u, v)
    for y, q in zip(fused, mods):
        assert same(y, q(x))
    two = ops.quantized_matmul_multi(x.reshape(1, m, k), mods[:2])
    assert two[0].shape == (1, m, 2048) and same(two[1].reshape(m, -1), mods[1](x))
