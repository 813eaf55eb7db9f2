"""GPU: the streamed decode GEMV (gbx_stream_encode / gbx_stream_run, csrc/gemv_stream.hip: per-wave LDS-DMA ring running
ahead of the activation prologue and of later ops) against (a) the launch-per-op fused GEMV, bit for bit where the
arithmetic order is identical (everything but the fused RMSNorm's fp32 sum), and (b) the CPU oracle with the stated
tolerance.  Reference op: mx.quantized_matmul at gbx_lm/models/quantized_linear_gba.py:195-203, M = 1."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.experimental]
DEV = "cuda:0"


def _same_bits(a, b, what=""):
    """Bit-for-bit comparison that says WHERE two results differ when they do (no second attempt: a mismatch fails the run).
    Round 5 wrapped these tests in a retry after ONE mismatch in one full-suite run that never reproduced (3 x 47 passes of this file alone,
    then three logged full-suite passes in round 6: profiles/r06_full_suite_passes.txt); the wrapper is gone, the diagnostics stay."""
    if torch.equal(a, b):
        return
    af, bf = a.float().reshape(-1).cpu(), b.float().reshape(-1).cpu()
    bad = torch.nonzero(af != bf).reshape(-1)
    idx = bad[:16].tolist()
    raise AssertionError(f"{what}: {bad.numel()} of {af.numel()} elements differ; first indices {idx}; values {af[bad[:8]].tolist()} vs "
                         f"{bf[bad[:8]].tolist()}; max |diff| {(af - bf).abs().max().item()}; "
                         f"index spacing {[int(bad[i + 1] - bad[i]) for i in range(min(8, bad.numel() - 1))]}")


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


def _pack(ops, mods, swiglu=False):
    if swiglu:
        return ops.pack_scales_zeros(mods[0], mods[1])
    return torch.cat([ops.pack_scales_zeros(m) for m in mods])


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("n,k,g", [(512, 4096, 128), (4096, 14336, 128), (1026, 3072, 64), (4096, 8192, 128), (66, 512, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_stream_single_matches_launch_path(oracle, bits, n, k, g, dtype):
    """one matrix (+ residual): bit-identical to gbx_qlinear_fused_forward, and within tolerance of the oracle"""
    from gbx_lm_amd import ops
    if bits == 2 and g == 32:
        pytest.skip("2-bit g32 stays on the launch path")
    m, d = _mk(oracle, n, k, bits, g, seed=bits * 7 + n)
    for t in ("scales", "zeros"):
        m._buffers[t] = m._buffers[t].to(dtype)
    x = d["x"].to(dtype).to(DEV).reshape(-1)
    h0 = torch.randn(n).to(dtype).to(DEV)
    pk = _pack(ops, (m,))
    y_old, y_new = h0.clone(), h0.clone()
    ops.qlinear_fused(ops.make_fused_desc(x, (m,), (y_old,), residual=y_old, packed=pk), x.device)
    prog = ops.StreamProgram([ops.make_fused_desc(x, (m,), (y_new,), residual=y_new, packed=pk)], DEV, dtype)
    prog.run()
    torch.cuda.synchronize()
    _same_bits(y_old, y_new, f"stream vs launch path, b{bits} n{n} k{k} g{g}")
    y64 = oracle.qlinear_ref(x.cpu()[None], d["qweight"], m.scales.cpu(), m.zeros.cpu(), bits, g, acc=np.float64)
    oracle.assert_close(y_new[None], h0.cpu().double().numpy()[None] + y64, "bf16" if dtype == torch.bfloat16 else "fp16",
                        f"stream residual b{bits}")


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("k", [4096, 3072])
def test_stream_norm_qkv(oracle, bits, k):
    from gbx_lm_amd import ops
    g = 128
    mods, ds = zip(*[_mk(oracle, n, k, bits, g, seed=10 * bits + i) for i, n in enumerate((4096, 1024, 1024))])
    x = (torch.randn(k) * 3).to(torch.bfloat16).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(torch.bfloat16).to(DEV)
    ys = [torch.zeros(m.output_dims, dtype=torch.bfloat16, device=DEV) for m in mods]
    pk = _pack(ops, mods)
    prog = ops.StreamProgram([ops.make_fused_desc(x, mods, ys, norm_weight=nw, eps=1e-5, packed=pk)], DEV, torch.bfloat16)
    prog.run()
    # without the norm the three segments must be bit-identical to the launch path
    ys0 = [torch.zeros_like(y) for y in ys]
    ys1 = [torch.zeros_like(y) for y in ys]
    ops.qlinear_fused(ops.make_fused_desc(x, mods, ys0, packed=pk), x.device)
    ops.StreamProgram([ops.make_fused_desc(x, mods, ys1, packed=pk)], DEV, torch.bfloat16).run()
    torch.cuda.synchronize()
    for a, b in zip(ys0, ys1):
        _same_bits(a, b, "stream vs launch path")
    xn = _rms(x, nw, 1e-5).cpu()
    for m, d, y in zip(mods, ds, ys):
        y64 = oracle.qlinear_ref(xn[None], d["qweight"], d["scales"], d["zeros"], bits, g, acc=np.float64)
        oracle.assert_close(y[None], y64, "bf16", f"stream qkv b{bits} K{k}")


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("n,k", [(14336, 4096), (1030, 4096), (8192, 3072)])
def test_stream_swiglu(oracle, bits, n, k):
    from gbx_lm_amd import ops
    g = 128
    (mg, dg), (mu, du) = _mk(oracle, n, k, bits, g, seed=bits), _mk(oracle, n, k, bits, g, seed=bits + 50)
    x = torch.randn(k).to(torch.bfloat16).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(torch.bfloat16).to(DEV)
    pk = _pack(ops, (mg, mu), swiglu=True)
    a_old = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    a_new = torch.zeros_like(a_old)
    ops.qlinear_fused(ops.make_fused_desc(x, (mg, mu), (a_old,), swiglu=True, packed=pk), x.device)
    ops.StreamProgram([ops.make_fused_desc(x, (mg, mu), (a_new,), swiglu=True, packed=pk)], DEV, torch.bfloat16).run()
    torch.cuda.synchronize()
    _same_bits(a_old, a_new, "stream vs launch path")
    # with the fused norm: against the oracle
    a_n = torch.zeros_like(a_old)
    ops.StreamProgram([ops.make_fused_desc(x, (mg, mu), (a_n,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pk)], DEV,
                      torch.bfloat16).run()
    torch.cuda.synchronize()
    xn = _rms(x, nw, 1e-5).cpu()
    g64 = oracle.qlinear_ref(xn[None], dg["qweight"], dg["scales"], dg["zeros"], bits, g, acc=np.float64)
    u64 = oracle.qlinear_ref(xn[None], du["qweight"], du["scales"], du["zeros"], bits, g, acc=np.float64)
    oracle.assert_close(a_n[None], (g64 / (1 + np.exp(-g64))) * u64, "bf16", f"stream swiglu b{bits}")


def test_stream_multi_op_program(oracle):
    """four independent ops of mixed widths in ONE launch: the issue cursor crosses ops (rows of op i+1 are requested while
    op i is consumed); every output bit-identical to its own launch"""
    from gbx_lm_amd import ops
    specs = [(2, (4096, 1024, 1024), 4096, False), (3, (4096,), 4096, False), (4, (2048, 2048), 4096, True), (2, (4096,), 14336, False),
             (4, (512,), 8192, False), (3, (1024, 256), 3072, False)]
    descs_new, descs_old, outs_new, outs_old, keep = [], [], [], [], []
    for si, (bits, ns, k, sw) in enumerate(specs):
        mods = [_mk(oracle, n, k, bits, 128, seed=100 + 10 * si + i)[0] for i, n in enumerate(ns)]
        x = torch.randn(k).to(torch.bfloat16).to(DEV)
        pk = _pack(ops, mods, swiglu=sw)
        nout = 1 if sw else len(ns)
        yo = [torch.zeros(ns[i], dtype=torch.bfloat16, device=DEV) for i in range(nout)]
        yn = [torch.zeros_like(t) for t in yo]
        descs_old.append(ops.make_fused_desc(x, mods, yo, swiglu=sw, packed=pk))
        descs_new.append(ops.make_fused_desc(x, mods, yn, swiglu=sw, packed=pk))
        outs_old.append(yo), outs_new.append(yn), keep.append((mods, x, pk))
    for d in descs_old:
        ops.qlinear_fused(d, torch.device(DEV))
    prog = ops.StreamProgram(descs_new, DEV, torch.bfloat16)
    prog.run()
    torch.cuda.synchronize()
    for si, (a, b) in enumerate(zip(outs_old, outs_new)):
        for u, v in zip(a, b):
            _same_bits(u, v, f"op {si}")
    # a sub-range of the same program
    for yn in outs_new:
        for t in yn:
            t.zero_()
    prog.run(2, 5)
    torch.cuda.synchronize()
    for si in (2, 3, 4):
        for u, v in zip(outs_old[si], outs_new[si]):
            _same_bits(u, v, f"sub-range op {si}")
    assert float(outs_new[0][0].abs().max()) == 0.0 and float(outs_new[5][0].abs().max()) == 0.0


def test_stream_encode_rejects(oracle):
    from gbx_lm_amd import ops
    m, _ = _mk(oracle, 64, 512, 2, 32, seed=1)
    x = torch.randn(512).to(torch.bfloat16).to(DEV)
    y = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises((ValueError, RuntimeError)):
        ops.StreamProgram([ops.make_fused_desc(x, (m,), (y,), packed=None)], DEV, torch.bfloat16)       # no packed companion
    with pytest.raises((ValueError, RuntimeError)):
        ops.StreamProgram([ops.make_fused_desc(x, (m,), (y,), packed=ops.pack_scales_zeros(m))], DEV, torch.bfloat16)   # 2-bit g32
