"""GPU: parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle would take minutes
per matrix): (1) the streaming GEMV / MFMA GEMM against a fp32 matmul with the SAME weights dequantised by
`gbx_dequantize` (itself bit-exact against the oracle at small sizes, test_qlinear_gpu.py); (2) linearity in x;
(3) column-sum checksum.  Shapes: every QuantizedLinear of Llama-3-8B and Llama-3.2-3B."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES_8B = [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)]
SHAPES_3B = [(3072, 3072), (1024, 3072), (8192, 3072), (3072, 8192)]


def _layer(n, k, bits, g, seed):
    gen = torch.Generator(device=DEV)
    gen.manual_seed(seed)
    qw = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=DEV, generator=gen).to(torch.int32).view(torch.uint32)
    qmax = float((1 << bits) - 1)
    sc = ((torch.rand((n, k // g), device=DEV, generator=gen) + 0.5) * (2.0 / qmax) * 0.02)
    ze = -sc * qmax / 2 + torch.randn((n, k // g), device=DEV, generator=gen) * 0.1 * sc
    return qw, sc.to(torch.bfloat16), ze.to(torch.bfloat16)


def _close(y, ref64, what):
    rms = ref64.pow(2).mean().sqrt().item()
    err = (y.double() - ref64).abs()
    bound = 2e-2 * rms + 2e-2 * ref64.abs()           # the stated bf16 tolerance (oracle.tolerance)
    assert (err <= bound).all(), f"{what}: {(err > bound).sum().item()} outside tolerance, max err {err.max().item():.4g} rms {rms:.4g}"


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("n,k", SHAPES_8B + SHAPES_3B)
def test_gemv_vs_dequantized_matmul(bits, n, k):
    from gbx_lm_amd import ops
    for g in (128, 32):
        qw, sc, ze = _layer(n, k, bits, g, seed=n + k + bits + g)
        x = torch.randn((1, k), device=DEV).to(torch.bfloat16)
        y = ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=bits)
        w = ops.dequantize(qw, sc, ze, group_size=g, bits=bits, dtype=torch.float32)
        ref = x.double() @ w.double().t()
        _close(y, ref, f"gemv N{n} K{k} b{bits} g{g}")
        # checksum of checksums: sum_n y[n] == x . (sum_n w[n, :]) -- catches dropped / duplicated rows
        colsum = w.double().sum(0)
        assert abs(y.double().sum().item() - (x.double() @ colsum).item()) <= 0.02 * ref.abs().sum().item() / np.sqrt(n) + 0.5


@pytest.mark.parametrize("bits", [2, 4])
def test_gemv_linearity_full_size(bits):
    from gbx_lm_amd import ops
    n, k, g = 14336, 4096, 128
    qw, sc, ze = _layer(n, k, bits, g, seed=77)
    x1 = torch.randn((1, k), device=DEV).to(torch.bfloat16)
    x2 = torch.randn((1, k), device=DEV).to(torch.bfloat16)
    f = lambda x: ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=bits).double()
    # 2*x1 and x1 + x2 (rounded to bf16 once, fed as the kernel sees them)
    x12 = (x1.float() + x2.float()).to(torch.bfloat16)
    exact = x12.double() - (x1.double() + x2.double())                 # rounding of the sum itself
    w = ops.dequantize(qw, sc, ze, group_size=g, bits=bits, dtype=torch.float32).double()
    lhs = f(x12)
    rhs = f(x1) + f(x2) + exact @ w.t()
    rms = lhs.pow(2).mean().sqrt().item()
    assert (lhs - rhs).abs().max().item() <= 0.03 * rms + 0.03 * lhs.abs().max().item() * 0.05
    assert torch.equal(f(x1 * 2), f(x1) * 2)                           # scaling by 2 is exact in every rounding step


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("m", [7, 64, 300])
def test_mfma_gemm_vs_dequantized_matmul(bits, m):
    from gbx_lm_amd import ops
    for (n, k) in [(4096, 4096), (1024, 14336), (1000, 1792)]:
        g = 128 if (n + m) % 2 == 0 else 64
        qw, sc, ze = _layer(n, k, bits, g, seed=m + bits)
        x = torch.randn((m, k), device=DEV).to(torch.bfloat16)
        y = ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=bits)
        w = ops.dequantize(qw, sc, ze, group_size=g, bits=bits, dtype=torch.float32)
        _close(y, x.double() @ w.double().t(), f"mfma M{m} N{n} K{k} b{bits} g{g}")


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_batch_kernel_full_size_rows_equal_single_token_gemv(bits):
    """cfg5 at full size (B = 32 rows, every 8B projection shape, g = 128): each row of the no-dequant matrix-core GEMM
    (mfma_batch.hip) against (1) fp64 x . dequantised weights and (2) the streaming GEMV of that row alone -- two
    kernels with unrelated arithmetic orders must agree within the output-dtype tolerance."""
    from gbx_lm_amd import ops
    for (n, k) in [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336)]:
        qw, sc, ze = _layer(n, k, bits, 128, seed=n + bits)
        x = torch.randn((32, k), device=DEV).to(torch.bfloat16)
        x[3, 100] = 60.0                                    # an outlier row entry
        y = ops.quantized_matmul(x, qw, sc, ze, group_size=128, bits=bits)
        w = ops.dequantize(qw, sc, ze, group_size=128, bits=bits, dtype=torch.float32)
        _close(y, x.double() @ w.double().t(), f"batch N{n} K{k} b{bits}")
        for i in (0, 3, 31):
            yi = ops.quantized_matmul(x[i:i + 1], qw, sc, ze, group_size=128, bits=bits)
            assert torch.allclose(y[i:i + 1].float(), yi.float(), rtol=2e-2, atol=2e-2 * yi.float().pow(2).mean().sqrt().item())


def test_fused_qkv_and_swiglu_full_size():
    """Fused entry points at the 8B block shapes against the unfused op + torch glue."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    def mod(n, k, bits, g, seed):
        m = QuantizedLinear(k, n, False, g, bits)
        qw, sc, ze = _layer(n, k, bits, g, seed)
        m._buffers["qweight"], m._buffers["scales"], m._buffers["zeros"] = qw, sc, ze
        return m
    k = 4096
    x = (torch.randn(k, device=DEV) * 2).to(torch.bfloat16)
    nw = (1 + 0.1 * torch.randn(k, device=DEV)).to(torch.bfloat16)
    xf = x.float()
    xn = (xf * torch.rsqrt(xf.pow(2).mean() + 1e-5) * nw.float()).to(torch.bfloat16)
    for bits in (2, 3, 4):
        q, kk, v = mod(4096, k, bits, 128, 1), mod(1024, k, bits, 128, 2), mod(1024, k, bits, 128, 3)
        ys = [torch.zeros(m.output_dims, dtype=torch.bfloat16, device=DEV) for m in (q, kk, v)]
        ops.qlinear_fused(ops.make_fused_desc(x, (q, kk, v), ys, norm_weight=nw, eps=1e-5), x.device)
        for m, y in zip((q, kk, v), ys):
            w = ops.dequantize(m.qweight, m.scales, m.zeros, group_size=128, bits=bits, dtype=torch.float32)
            _close(y[None], xn[None].double() @ w.double().t(), f"fused qkv b{bits}")
        gate, up = mod(14336, k, bits, 128, 4), mod(14336, k, bits, 128, 5)
        act = torch.zeros(14336, dtype=torch.bfloat16, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (gate, up), (act,), norm_weight=nw, eps=1e-5, swiglu=True), x.device)
        wg = ops.dequantize(gate.qweight, gate.scales, gate.zeros, group_size=128, bits=bits, dtype=torch.float32).double()
        wu = ops.dequantize(up.qweight, up.scales, up.zeros, group_size=128, bits=bits, dtype=torch.float32).double()
        gg, uu = xn[None].double() @ wg.t(), xn[None].double() @ wu.t()
        _close(act[None], gg / (1 + torch.exp(-gg)) * uu, f"fused swiglu b{bits}")


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("n,k", SHAPES_8B)
def test_dequantize_full_size_rows_vs_c_oracle(c_oracle, bits, n, k):
    """Chain of trust at the sizes it is used: gbx_dequantize (the reference weights of every full-size test here and in
    test_prefill_gpu.py) against the C oracle's dequant on a strided subset of rows, bit for bit (fp32 fma of exact
    integers and 16-bit scale/zero: one correctly rounded result), for g = 128 and 64."""
    from gbx_lm_amd import ops
    for g in (128, 64):
        qw, sc, ze = _layer(n, k, bits, g, seed=3 * n + k + bits + g)
        w = ops.dequantize(qw, sc, ze, group_size=g, bits=bits, dtype=torch.float32)
        rows = sorted({0, 1, n - 1, n // 2, *range(3, n, max(n // 29, 1))})
        ref = c_oracle.dequant(qw.view(torch.int32)[rows].cpu().numpy().view(np.uint32), sc[rows].cpu(), ze[rows].cpu(), bits, g)
        got = w[rows].cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), \
            f"dequantize N{n} K{k} b{bits} g{g}: {int((got != ref).sum())} elements differ"
