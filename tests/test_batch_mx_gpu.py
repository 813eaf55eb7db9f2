"""GPU, experimental library: the M <= 32 batch GEMM on the FP4 x FP8 matrix cores (gbx_exp_batch_mx_image / _forward, csrc/batch_mx.hip)
against fp64 on dequantised weights (oracle), with the two-limb tolerance of the decode kernel of the same arithmetic, and against the
product batch GEMM.  Reference op: mx.quantized_matmul at x [B, 1, K] (gbx_lm/models/quantized_linear_gba.py:195-203)."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.experimental]
DEV = "cuda:0"


def _mk(oracle, n, k, seed):
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    d = oracle.synth_layer(n, k, 2, 128, seed=seed)
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m, d


def _run(x, mod):
    from gbx_lm_amd import _lib
    L = _lib.exp_lib()
    m, k = x.shape
    n = mod.output_dims
    img = torch.empty(L.gbx_exp_batch_mx_image_bytes(k), dtype=torch.uint8, device=DEV)
    aux = torch.empty(L.gbx_exp_batch_mx_aux_bytes(k), dtype=torch.uint8, device=DEV)
    y = torch.zeros(m, n, dtype=torch.bfloat16, device=DEV)
    ks = L.gbx_exp_batch_mx_slices(n, k)
    assert ks >= 1, "shape not covered"
    ws = torch.zeros(ks, m, n, dtype=torch.float32, device=DEV) if ks > 1 else None
    _lib.check_exp(L.gbx_exp_batch_mx_image(x.data_ptr(), m, k, img.data_ptr(), aux.data_ptr(), None))
    _lib.check_exp(L.gbx_exp_batch_mx_forward(img.data_ptr(), aux.data_ptr(), mod.qweight.data_ptr(), mod.scales.data_ptr(), mod.zeros.data_ptr(),
                                              y.data_ptr(), ws.data_ptr() if ws is not None else None, m, n, k, None))
    if ws is not None:                      # k split across workgroups: the consumer sums the slices (one rounding)
        acc = ws[0].clone()
        for i in range(1, ks):
            acc += ws[i]
        y = acc.to(torch.bfloat16)
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("m,n,k", [(32, 256, 1024), (32, 4096, 4096), (5, 1024, 2048), (17, 512, 14336), (1, 128, 1024), (32, 28672, 4096), (32, 4096, 14336)])
def test_batch_mx_vs_fp64(oracle, m, n, k):
    mod, d = _mk(oracle, n, k, seed=m + n)
    x = (torch.randn(m, k) * 0.8).to(torch.bfloat16).to(DEV)
    y = _run(x, mod)
    y64 = oracle.qlinear_ref(x.cpu(), d["qweight"], d["scales"], d["zeros"], 2, 128, acc=np.float64)
    oracle.assert_close(y, y64, "bf16", f"batch mx M{m} N{n} K{k}")
    # and the product batch GEMM on the same inputs: both within the tolerance of the same oracle, close to each other
    y_prod = mod(x)
    err = (y.float() - y_prod.float()).abs()
    assert (err <= 2e-2 * y_prod.float().abs() + 2e-2 * y_prod.float().pow(2).mean().sqrt()).all()


def test_batch_mx_one_hot_rows_are_exact(oracle):
    """x = one-hot rows (a single 1.0 per row at varied positions of the first / last groups): y[m, n] = w[n, k_m] exactly (one limb
    carries 1.0, every other product is zero) -- pins the operand slot map (which k an image byte stands for) at every k-block."""
    n, k = 256, 1024
    mod, d = _mk(oracle, n, k, seed=3)
    from gbx_lm_amd import ops
    w = ops.dequantize(mod.qweight, mod.scales, mod.zeros, 128, 2, dtype=torch.float32)          # [n, k]
    pos = [0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 95, 96, 127, 128, 129, 255, 256, 300, 383, 384, 385, 400, 447, 448, 470, 495, 496, 509, 510, 1023, 7]
    x = torch.zeros(32, k, dtype=torch.bfloat16, device=DEV)
    for m_, p in enumerate(pos):
        x[m_, p] = 1.0
    y = _run(x, mod)
    want = torch.stack([w[:, p] for p in pos]).to(torch.bfloat16)
    assert torch.equal(y, want), f"{int((y != want).sum())} of {y.numel()} differ"
