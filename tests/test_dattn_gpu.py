"""GPU: the fused attention launch of round 5 (gbx_dattn_forward: RMSNorm -> q|k|v -> RoPE + cache append -> attention -> o_proj
partial sums, one kv head per XCD) and its two companions (gbx_dgemv.x_acc / x_out prologue, zero_buf), through the C-ABI.

Reference: the same block computed in fp64 on gbx_dequantize weights with the reference graph's roundings
(gbx_lm/models/qllama.py:68-96,131-141; mx.quantized_matmul at M == 1, quantized_linear_gba.py:195-203) -- the tolerance is the
decode GEMV's (2e-2 of the value + 2e-2 of the vector's RMS), stated at each assert.  Integer / index work (the cache row the
launch appends, the accumulator's fixed-point format, the generation tags over repeated launches) is checked exactly.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DT = torch.bfloat16           # activation dtype of the module's helpers (the fp16 test below swaps it)


def _bf(x):
    return x.to(DT).double()


def _deq(m):
    from gbx_lm_amd import ops
    return ops.dequantize(m.qweight, m.scales, m.zeros, m.group_size, m.bits, dtype=torch.float32).double()


def _layer(cfg_name, bits_qkv, bits_o, group=128, seed=0):
    from gbx_lm_amd import synthetic as S
    cfg = dict(getattr(S, cfg_name), num_hidden_layers=1, vocab_size=512)
    strat = S.make_strategy("S22", 1, group)
    for p in strat["model.layers.0"]:
        b = bits_o if p.endswith("o_proj") else bits_qkv
        strat["model.layers.0"][p] = {"bits": [b], "group_size": {str(b): group}}
    return S.build_model(cfg, strat, device=DEV, seed=seed, act_dtype=DT), cfg


def _cat_qkv(at):
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    mods = (at.q_proj, at.k_proj, at.v_proj)
    cat = QuantizedLinear(at.q_proj.input_dims, sum(m.output_dims for m in mods), False, at.q_proj.group_size, at.q_proj.bits)
    for nm in ("qweight", "scales", "zeros"):
        cat._buffers[nm] = torch.cat([getattr(m, nm) for m in mods]).contiguous()
    return cat


def _reference(blk, h, pos, kc, vc):
    """fp64 on dequantised weights, bf16 roundings where the reference graph holds a bf16 tensor.  Returns (q_rot, k_rot, v, attn, o)."""
    from gbx_lm_amd.models.rope_utils import apply_rope
    at = blk.self_attn
    hd, nh, nkv = at.head_dim, at.n_heads, at.n_kv_heads
    xf = h.double()
    xn = _bf(xf * torch.rsqrt(xf.pow(2).mean() + blk.input_layernorm.eps) * blk.input_layernorm.weight.double())
    q, k, v = _bf(_deq(at.q_proj) @ xn), _bf(_deq(at.k_proj) @ xn), _bf(_deq(at.v_proj) @ xn)
    inv = at.rope.inv_freq.cpu()
    sc = getattr(at.rope, "scale", 1.0)
    qr = apply_rope(q.to(DT).cpu().view(1, nh, 1, hd), inv, pos, sc, at.rope.traditional).to(DEV).double().view(nh, hd)
    kr = apply_rope(k.to(DT).cpu().view(1, nkv, 1, hd), inv, pos, sc, at.rope.traditional).to(DEV).double().view(nkv, hd)
    K = torch.cat([kc[:, :pos].double(), kr[:, None]], 1)
    V = torch.cat([vc[:, :pos].double(), v.view(nkv, 1, hd)], 1)
    rep = nh // nkv
    s = torch.einsum("hd,htd->ht", qr, K.repeat_interleave(rep, 0)) * at.scale
    attn = _bf(torch.einsum("ht,htd->hd", torch.softmax(s, -1), V.repeat_interleave(rep, 0)).reshape(-1))
    return kr, v.view(nkv, hd), attn


def _close(got, want, what, rel=2e-2):
    err = (got.double() - want).abs()
    tol = rel * want.abs() + rel * want.pow(2).mean().sqrt()
    assert (err <= tol).all(), f"{what}: {int((err > tol).sum())} of {want.numel()} outside tolerance, worst {err.max().item():.4g}"


CASES = [
    ("LLAMA3_8B", 2, 2, 40), ("LLAMA3_8B", 2, 2, 300), ("LLAMA3_8B", 3, 3, 9), ("LLAMA3_8B", 4, 4, 131), ("LLAMA3_8B", 2, 4, 200), ("LLAMA3_8B", 4, 2, 0),
    ("LLAMA32_3B", 2, 2, 17), ("LLAMA32_3B", 4, 4, 130), ("LLAMA32_3B", 3, 2, 64),
    # beyond GBX_DATTN_COOP_MIN (384) positions the group's producers sweep the context in chunks (cooperative sweep): the first
    # context that takes it, chunks that leave whole waves idle, ragged last chunks, a chunk longer than the register ring (beyond
    # ~3.5 K positions), three query heads per kv head (the second head pair holds one head)
    ("LLAMA3_8B", 2, 2, 385), ("LLAMA3_8B", 2, 2, 1500), ("LLAMA3_8B", 3, 4, 1023), ("LLAMA3_8B", 2, 2, 4000), ("LLAMA3_8B", 4, 2, 449),
    ("LLAMA32_3B", 2, 2, 700), ("LLAMA32_3B", 4, 3, 2047), ("LLAMA32_3B", 3, 3, 390),
]


@pytest.mark.parametrize("cfg_name,bq,bo,ctx", CASES)
def test_dattn_launch_vs_fp64(cfg_name, bq, bo, ctx):
    from gbx_lm_amd import ops
    model, cfg = _layer(cfg_name, bq, bo, seed=bq * 10 + bo)
    blk = model.model.layers[0]
    at = blk.self_attn
    D, nh, nkv, hd = cfg["hidden_size"], at.n_heads, at.n_kv_heads, at.head_dim
    cat = _cat_qkv(at)
    assert ops.dattn_supported(D, nh, nkv, hd, cat, at.o_proj), "the BASELINE geometries must be covered"
    cap = 256
    while cap <= ctx:
        cap *= 2
    gen = torch.Generator(device="cpu").manual_seed(ctx + 1)
    h = (torch.randn(D, generator=gen) * 0.7).to(DT).to(DEV)
    kc = (torch.randn(1, nkv, cap, hd, generator=gen) * 0.5).to(DT).to(DEV)
    vc = (torch.randn(1, nkv, cap, hd, generator=gen)).to(DT).to(DEV)
    kc0, vc0 = kc.clone(), vc.clone()
    pos = torch.tensor([ctx], dtype=torch.long, device=DEV)
    acc = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
    state = ops.dattn_state(DEV)
    attn_out = torch.zeros(nh * hd, dtype=DT, device=DEV)
    pk = ops.dgemv_pack(cat, 2)
    d = ops.make_dattn_desc(h, blk.input_layernorm.weight, blk.input_layernorm.eps, cat, pk, kc, vc, at.rope.inv_freq, pos, at.o_proj,
                            acc, state, nh, cap, getattr(at.rope, "scale", 1.0), at.scale, at.rope.traditional, attn_out=attn_out)
    ops.dattn(d, DEV)
    torch.cuda.synchronize()
    assert ops.dattn_status(state) == 0, "a hand-off timed out"
    kr, v, attn = _reference(blk, h, ctx, kc0[0], vc0[0])
    # the appended cache row: RoPE(k) and v of the new token, everything else untouched (exact)
    _close(kc[0, :, ctx], kr, "appended K row", rel=1e-2)
    _close(vc[0, :, ctx], v, "appended V row", rel=2e-2)
    keep = torch.ones(cap, dtype=torch.bool, device=DEV)
    keep[ctx] = False
    assert torch.equal(kc[0][:, keep], kc0[0][:, keep]) and torch.equal(vc[0][:, keep], vc0[0][:, keep]), "rows other than `pos` changed"
    _close(attn_out, attn, "attention output")
    # o_proj: against the launch's OWN attention output (isolates phase 3), fixed point -> float
    o_got = acc[:D].double() * 2.0 ** -32
    o_want = _deq(at.o_proj) @ attn_out.double()
    _close(o_got, o_want, "o_proj partial sums", rel=1e-2)
    assert int(acc[D].item()) == 0, "non-finite marker set on finite data"


def test_dattn_fp16_activations():
    """The launch's fp16 instantiations (vector-ALU form; the matrix-core form is bf16 only): short context, the cooperative sweep,
    3- and 4-bit rows."""
    global DT
    DT = torch.float16
    try:
        for case in (("LLAMA3_8B", 2, 2, 40), ("LLAMA3_8B", 2, 2, 500), ("LLAMA3_8B", 3, 4, 131), ("LLAMA32_3B", 4, 2, 450)):
            test_dattn_launch_vs_fp64(*case)
    finally:
        DT = torch.bfloat16


def test_dattn_vector_form_when_the_matrix_core_form_is_switched_off():
    """GBX_DATTN_MX=0 (read once per process: a child): the 2-bit 8B launch on the vector-ALU dots, same oracle check."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_dattn_gpu as t\n"
            "t.test_dattn_launch_vs_fp64('LLAMA3_8B', 2, 2, 40)\n"
            "t.test_dattn_launch_vs_fp64('LLAMA3_8B', 2, 2, 170)\n"
            "print('VEC_OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GBX_DATTN_MX="0"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "VEC_OK" in out.stdout, out.stderr[-2000:]


def test_dattn_cooperative_sweep_at_short_contexts_and_on_the_vector_form():
    """GBX_DATTN_COOP_MIN=8 (a child: read once per process) sends short contexts through the cooperative sweep -- chunks of four
    positions, most producers with an empty chunk -- and GBX_DATTN_MX=0 puts the producers' phase 1 on the vector-ALU form before it."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_dattn_gpu as t\n"
            "for case in (('LLAMA3_8B', 2, 2, 9), ('LLAMA3_8B', 2, 2, 40), ('LLAMA3_8B', 4, 3, 131), ('LLAMA32_3B', 2, 2, 17), ('LLAMA3_8B', 2, 2, 500)):\n"
            "    t.test_dattn_launch_vs_fp64(*case)\n"
            "print('COOP_OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    for mx in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GBX_DATTN_COOP_MIN="8", GBX_DATTN_MX=mx),
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "COOP_OK" in out.stdout, out.stderr[-2000:]


def test_dattn_repeated_launches_and_accumulator_protocol():
    """Generation tags: 50 launches back to back on one state buffer (each must consume the hand-offs of ITS launch: the inputs change
    every time); the accumulator ADDS (two launches without clearing = twice the sums, integer-exact); results are bit-reproducible."""
    from gbx_lm_amd import ops
    model, cfg = _layer("LLAMA3_8B", 2, 2, seed=3)
    blk = model.model.layers[0]
    at = blk.self_attn
    D, nh, nkv, hd = cfg["hidden_size"], at.n_heads, at.n_kv_heads, at.head_dim
    cat = _cat_qkv(at)
    cap = 256
    gen = torch.Generator(device="cpu").manual_seed(5)
    hs = (torch.randn(50, D, generator=gen)).to(torch.bfloat16).to(DEV)
    h = hs[0].clone()
    kc = torch.zeros(1, nkv, cap, hd, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    pos = torch.zeros(1, dtype=torch.long, device=DEV)
    acc = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
    state = ops.dattn_state(DEV)
    attn_out = torch.zeros(nh * hd, dtype=torch.bfloat16, device=DEV)
    pk = ops.dgemv_pack(cat, 2)
    d = ops.make_dattn_desc(h, blk.input_layernorm.weight, blk.input_layernorm.eps, cat, pk, kc, vc, at.rope.inv_freq, pos, at.o_proj,
                            acc, state, nh, cap, getattr(at.rope, "scale", 1.0), at.scale, at.rope.traditional, attn_out=attn_out)
    outs = []
    for t in range(50):
        h.copy_(hs[t])
        pos.fill_(t)
        acc.zero_()
        ops.dattn(d, DEV)
        outs.append((attn_out.clone(), acc.clone()))
    torch.cuda.synchronize()
    assert ops.dattn_status(state) == 0
    # replay the last step from a copy of the cache as it was: identical bits (order-independent fixed-point sums)
    kc2, vc2 = kc.clone(), vc.clone()
    acc.zero_()
    ops.dattn(d, DEV)
    assert torch.equal(attn_out, outs[-1][0]) and torch.equal(acc, outs[-1][1]), "the launch is not bit-reproducible"
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)
    ops.dattn(d, DEV)                                          # no clearing: the sums add
    assert torch.equal(acc[:D], 2 * outs[-1][1][:D])
    # step t against the fp64 reference on the cache the earlier steps built
    kr, v, attn = _reference(blk, hs[49], 49, kc2[0], vc2[0])
    _close(outs[-1][0], attn, "attention output of step 49")


@pytest.mark.parametrize("bits", [2, 4])
def test_gateup_prologue_reads_the_accumulator(bits):
    """gbx_dgemv with x_acc / x_out: x' = round(x + round(acc * 2^-32)) formed in the prologue must give the launch's output on x' and
    write x' once; zero_buf clears; the non-finite marker turns x' into NaN."""
    from gbx_lm_amd import _lib, ops
    model, cfg = _layer("LLAMA3_8B", bits, bits, seed=7)
    mlp = model.model.layers[0].mlp
    nw = model.model.layers[0].post_attention_layernorm
    D, I = cfg["hidden_size"], cfg["intermediate_size"]
    gen = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(D, generator=gen).to(torch.bfloat16).to(DEV)
    o = (torch.randn(D, generator=gen) * 0.3).double()
    acc = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
    acc[:D] = torch.round(o * 2.0 ** 32).to(torch.int64).to(DEV)
    xo = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros(I, dtype=torch.bfloat16, device=DEV)
    y2 = torch.zeros_like(y)
    pl = ops.dgemv_plan(I, D, bits, 128, _lib.GBX_DG_SWIGLU, True)
    pk = ops.dgemv_pack(mlp.gate_proj, pl[0], mlp.up_proj)
    d = ops.make_dgemv_desc(x, mlp.gate_proj, y, pk, pl[0], mod_up=mlp.up_proj, norm_weight=nw.weight, eps=nw.eps, x_acc=acc, x_out=xo)
    ops.dgemv(d, DEV)
    want_x = (x.double() + (acc[:D].double() * 2.0 ** -32).to(torch.bfloat16).double()).to(torch.bfloat16)
    assert torch.equal(xo, want_x), "x_out is not round(x + round(acc * 2^-32))"
    d2 = ops.make_dgemv_desc(want_x, mlp.gate_proj, y2, pk, pl[0], mod_up=mlp.up_proj, norm_weight=nw.weight, eps=nw.eps)
    ops.dgemv(d2, DEV)
    # same arithmetic on the same x'; the sum of squares is reduced in a different order (with x' instead of per item), so rstd may move
    # by an fp32 ulp and an output by one bf16 rounding step
    yf, y2f = y.float(), y2.float()
    assert ((yf - y2f).abs() <= 2.0 ** -7 * y2f.abs() + 1e-6).all() and (yf == y2f).float().mean() > 0.98, "the accumulator prologue changed the launch's arithmetic"
    # zero_buf on a residual launch
    pl_dn = ops.dgemv_plan(D, I, bits, 128, _lib.GBX_DG_RESIDUAL, False)
    pk_dn = ops.dgemv_pack(mlp.down_proj, pl_dn[0])
    hb = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
    d3 = ops.make_dgemv_desc(y, mlp.down_proj, hb, pk_dn, pl_dn[0], residual=xo, zero_buf=acc)
    ops.dgemv(d3, DEV)
    assert int(acc.abs().sum().item()) == 0, "zero_buf left words behind"
    d4 = ops.make_dgemv_desc(y, mlp.down_proj, torch.zeros_like(hb), pk_dn, pl_dn[0], residual=xo)
    ops.dgemv(d4, DEV)
    # non-finite marker
    acc[D] = 1
    ops.dgemv(d, DEV)
    assert torch.isnan(xo.float()).all() and torch.isnan(y.float()).all()
    with pytest.raises(ValueError):
        bad = ops.make_dgemv_desc(x, mlp.gate_proj, y, pk, pl[0], mod_up=mlp.up_proj, norm_weight=nw.weight, eps=nw.eps, x_acc=acc, x_out=x)
        ops.dgemv(bad, DEV)


@pytest.mark.parametrize("cfg_name,bits,group,seed", [("LLAMA3_8B", 2, 128, 0), ("LLAMA3_8B", 2, 128, 1),
                                                      # round 6: the vector-ALU form (dmlpv.hip) for 3- and 4-bit rows, both BASELINE geometries, every group size
                                                      ("LLAMA3_8B", 3, 128, 2), ("LLAMA3_8B", 4, 128, 3), ("LLAMA32_3B", 3, 128, 4), ("LLAMA32_3B", 4, 128, 5),
                                                      ("LLAMA3_8B", 4, 64, 6), ("LLAMA32_3B", 3, 32, 7), ("LLAMA3_8B", 3, 64, 8),
                                                      # (4 bits at group 128 now take the matrix-core form -- a nibble as two 2-bit fields, dmlp_kernel<2,4> -- at both
                                                      #  geometries; K = 3072 is one and a half spans of operands: the 2-bit kernel at that geometry too)
                                                      ("LLAMA32_3B", 2, 128, 9)])
def test_dmlp_launch_vs_fp64(cfg_name, bits, group, seed):
    """gbx_dmlp_forward (2 bits: FP4 x FP8 matrix cores, Llama-3-8B MLP shapes; 3 / 4 bits: the decode GEMV's vector-ALU arithmetic, 8B and
    3.2-3B shapes): x' exact; the SwiGLU activations against fp64 on dequantised weights with the reference graph's roundings (2e-2 of the
    value + 2e-2 of the RMS, as tests/test_dgemv_gpu.py); down_proj's fixed-point sums against fp64 on the launch's OWN activations;
    zero_buf; bit-reproducible."""
    from gbx_lm_amd import ops
    model, cfg = _layer(cfg_name, bits, bits, group=group, seed=20 + seed)
    blk = model.model.layers[0]
    mlp, nw = blk.mlp, blk.post_attention_layernorm
    D, I = cfg["hidden_size"], cfg["intermediate_size"]
    assert ops.dmlp_supported(D, mlp.gate_proj, mlp.up_proj, mlp.down_proj, torch.bfloat16)
    gen = torch.Generator(device="cpu").manual_seed(100 + seed)
    x = torch.randn(D, generator=gen).to(torch.bfloat16).to(DEV)
    o = (torch.randn(D, generator=gen) * 0.3).double()
    acc_in = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
    acc_in[:D] = torch.round(o * 2.0 ** 32).to(torch.int64).to(DEV)
    x_out = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
    acc_out = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
    junk = torch.full((D + 1,), 7, dtype=torch.int64, device=DEV)
    act = torch.zeros(I, dtype=torch.bfloat16, device=DEV)
    state = ops.dattn_state(DEV)
    pk_gu, pk_dn = ops.dgemv_pack(mlp.gate_proj, 2, mlp.up_proj), ops.dgemv_pack(mlp.down_proj, 2)
    d = ops.make_dmlp_desc(x, acc_in, x_out, nw.weight, nw.eps, mlp.gate_proj, mlp.up_proj, mlp.down_proj, pk_gu, pk_dn, acc_out, state,
                           zero_buf=junk, act_out=act)
    ops.dmlp(d, DEV)
    torch.cuda.synchronize()
    assert ops.dattn_status(state) == 0, "the hand-off timed out"
    want_x = (x.double() + (acc_in[:D].double() * 2.0 ** -32).to(torch.bfloat16).double()).to(torch.bfloat16)
    assert torch.equal(x_out, want_x), "x_out is not round(x + round(acc * 2^-32))"
    assert int(junk.abs().sum().item()) == 0, "zero_buf left words behind"
    xf = want_x.double()
    xn = _bf(xf * torch.rsqrt(xf.pow(2).mean() + nw.eps) * nw.weight.double())
    g, u = _bf(_deq(mlp.gate_proj) @ xn), _bf(_deq(mlp.up_proj) @ xn)
    want_act = _bf(_bf(g / (1 + torch.exp(-g))) * u)
    _close(act, want_act, "SwiGLU activations")
    got = acc_out[:D].double() * 2.0 ** -32
    want = _deq(mlp.down_proj) @ act.double()
    _close(got, want, "down_proj sums")
    assert int(acc_out[D].item()) == 0
    first = acc_out.clone()
    acc_out.zero_()
    ops.dmlp(d, DEV)                                            # a second launch on the same state: next generation, identical bits
    assert torch.equal(acc_out, first) and ops.dattn_status(state) == 0
    # the step after the last block: sums -> stream, accumulators cleared
    out = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
    z1 = torch.full((D + 1,), 3, dtype=torch.int64, device=DEV)
    ops.acc_residual(x_out, acc_out, out, zero=[z1])
    assert torch.equal(out, (x_out.double() + (first[:D].double() * 2.0 ** -32).to(torch.bfloat16).double()).to(torch.bfloat16))
    assert int(acc_out.abs().sum().item()) == 0 and int(z1.abs().sum().item()) == 0


def test_dmlp_zero_and_non_finite_inputs():
    """All-zero stream -> exact zero sums; a NaN / Inf anywhere in the stream -> the non-finite marker of acc_out is raised and the
    next consumer (gbx_acc_residual here) produces NaN: the matrix-core form's documented rule for non-finite activations."""
    from gbx_lm_amd import ops
    model, cfg = _layer("LLAMA3_8B", 2, 2, seed=41)
    blk = model.model.layers[0]
    mlp, nw = blk.mlp, blk.post_attention_layernorm
    D, I = cfg["hidden_size"], cfg["intermediate_size"]
    pk_gu, pk_dn = ops.dgemv_pack(mlp.gate_proj, 2, mlp.up_proj), ops.dgemv_pack(mlp.down_proj, 2)
    state = ops.dattn_state(DEV)
    for bad in (None, float("nan"), float("inf")):
        x = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
        if bad is not None:
            x = torch.randn(D).to(torch.bfloat16).to(DEV)
            x[D // 2 + 3] = bad
        acc_in = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
        x_out = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
        acc_out = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
        d = ops.make_dmlp_desc(x, acc_in, x_out, nw.weight, nw.eps, mlp.gate_proj, mlp.up_proj, mlp.down_proj, pk_gu, pk_dn, acc_out, state)
        ops.dmlp(d, DEV)
        torch.cuda.synchronize()
        assert ops.dattn_status(state) == 0
        if bad is None:
            assert int(acc_out.abs().sum().item()) == 0, "zero activations must give exactly zero sums"
        else:
            assert int(acc_out[D].item()) > 0, "non-finite marker not raised"
            out = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
            ops.acc_residual(x_out, acc_out, out)
            assert torch.isnan(out.float()).all()


def test_dattn_takes_the_stream_as_vector_plus_sums():
    """gbx_dattn.x_acc / x_out: the launch on (x, sums) equals the launch on the finished vector, writes it once and clears zero_buf."""
    from gbx_lm_amd import ops
    model, cfg = _layer("LLAMA3_8B", 2, 2, seed=31)
    blk = model.model.layers[0]
    at = blk.self_attn
    D, nh, nkv, hd = cfg["hidden_size"], at.n_heads, at.n_kv_heads, at.head_dim
    cat = _cat_qkv(at)
    cap, ctx = 256, 33
    gen = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn(D, generator=gen).to(torch.bfloat16).to(DEV)
    dsum = (torch.randn(D, generator=gen) * 0.5).double()
    xacc = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
    xacc[:D] = torch.round(dsum * 2.0 ** 32).to(torch.int64).to(DEV)
    h = (x.double() + (xacc[:D].double() * 2.0 ** -32).to(torch.bfloat16).double()).to(torch.bfloat16)
    kc = (torch.randn(1, nkv, cap, hd, generator=gen) * 0.5).to(torch.bfloat16).to(DEV)
    vc = torch.randn(1, nkv, cap, hd, generator=gen).to(torch.bfloat16).to(DEV)
    pos = torch.tensor([ctx], dtype=torch.long, device=DEV)
    pk = ops.dgemv_pack(cat, 2)
    outs = []
    for mode in (0, 1):
        acc = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
        state = ops.dattn_state(DEV)
        ao = torch.zeros(nh * hd, dtype=torch.bfloat16, device=DEV)
        k2, v2 = kc.clone(), vc.clone()
        xo = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
        junk = torch.full((D + 1,), 5, dtype=torch.int64, device=DEV)
        kw = dict(x_acc=xacc, x_out=xo, zero_buf=junk) if mode else {}
        d = ops.make_dattn_desc(x if mode else h, blk.input_layernorm.weight, blk.input_layernorm.eps, cat, pk, k2, v2, at.rope.inv_freq, pos,
                                at.o_proj, acc, state, nh, cap, getattr(at.rope, "scale", 1.0), at.scale, at.rope.traditional, attn_out=ao, **kw)
        ops.dattn(d, DEV)
        torch.cuda.synchronize()
        assert ops.dattn_status(state) == 0
        if mode:
            assert torch.equal(xo, h), "x_out is not round(x + round(x_acc * 2^-32))"
            assert int(junk.abs().sum().item()) == 0
        outs.append((ao, acc, k2, v2))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_), "the launch on (x, sums) differs from the launch on the finished vector"


def test_dattn_plan_refuses_what_it_does_not_cover():
    from gbx_lm_amd import _lib
    L = _lib.lib()
    ok = lambda *a: L.gbx_dattn_plan(*a) == _lib.GBX_OK
    assert ok(4096, 32, 8, 128, 2, 128, 2, 128) and ok(3072, 24, 8, 128, 4, 128, 4, 128) and ok(4096, 32, 8, 128, 3, 64, 4, 32)
    assert not ok(4096, 32, 4, 128, 2, 128, 2, 128), "4 kv heads: no XCD bijection"
    assert not ok(2048, 32, 8, 64, 4, 128, 4, 128), "head_dim 64"
    assert not ok(8192, 64, 8, 128, 2, 128, 2, 128), "rows longer than one lane pass at 2 bits"
    assert not ok(4096, 32, 8, 128, 2, 32, 2, 128), "2-bit group 32: two (scale, zero) pairs per lane slot"


def _hb(env_extra, layers=3, steps=12, strategy="S22", cfg_name="LLAMA3_8B"):
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "import bench\n"
        "from gbx_lm_amd import synthetic as S\n"
        "from gbx_lm_amd.decode import FusedDecodeRunner\n"
        "cfg = dict(S.%s, num_hidden_layers=%d, vocab_size=2048)\n"
        "model = S.build_model(cfg, S.make_strategy('%s', %d, 128), device='cuda:0', seed=5)\n"
        "caches = bench.make_caches(model, 1, 256, 'cuda:0')\n"
        "r = FusedDecodeRunner(model, caches, use_graph=True)\n"
        "print('DATTN', sum(d is not None for d in r.da))\n"
        "print('DMLP', sum(d is not None and len(d) == 2 for d in r.da))\n"
        "for t in range(%d): lg = r.step(torch.tensor([(7 * t + 1) %% 2048], device='cuda:0'))\n"
        "torch.cuda.synchronize()\n"
        "r.dattn_check()\n"
        "print('HB', r.hb.float().cpu().numpy().tobytes().hex())\n"
        "print('KC', caches[-1].keys[0, :, :%d].float().cpu().numpy().tobytes().hex())\n" % (ROOT, cfg_name, layers, strategy, layers, steps, steps))
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return {l.split()[0]: l.split()[1] for l in out.stdout.splitlines() if l.startswith(("HB", "DATTN", "DMLP", "KC"))}


def test_three_launch_block_agrees_with_five_launch_block():
    """Three 8B-shaped layers of mixed widths, 12 teacher-forced steps through the graph-replayed runner: GBX_DATTN=1 (three launches
    per block) against GBX_DATTN=0 (five).  The o_proj sum is formed in a different order (eight fixed-point partials instead of one
    fp32 chain), so agreement is to a few bf16 ulps of the stream, not to the bit."""
    new, old = _hb({"GBX_DATTN": "1", "GBX_DMLP": "0"}), _hb({"GBX_DATTN": "0"})
    assert int(new["DATTN"]) == 3 and int(old["DATTN"]) == 0 and int(new["DMLP"]) == 0
    f = lambda d, k: np.frombuffer(bytes.fromhex(d[k]), dtype=np.float32)
    a, b = f(new, "HB"), f(old, "HB")
    assert np.abs(a - b).max() <= 0.05 * np.abs(b).mean() + 0.05 * np.abs(b).max(), "residual stream after 12 steps"
    ka, kb = f(new, "KC"), f(old, "KC")
    assert np.abs(ka - kb).max() <= 0.05 * np.abs(kb).mean() + 0.05 * np.abs(kb).max(), "last layer's cached keys"


def test_two_launch_block_agrees_with_five_launch_block():
    """Five 8B-shaped layers, all 2 bits (S20) and the S22 mix on eight layers (3-bit blocks on three launches between 2- and 4-bit
    blocks on two): the stream passes between blocks as (vector, fixed-point sums), accumulators alternate by parity, the last
    block's sums are finished by gbx_acc_residual.  Agreement with the five-launch block to a few bf16 ulps of the stream."""
    f = lambda d, k: np.frombuffer(bytes.fromhex(d[k]), dtype=np.float32)
    # (round 6: in a mostly-4-bit model the 4-bit blocks take the fused MLP launch too -- not the last block; the 3.2-3B geometry, K = 3072, as well)
    # (mostly-3-bit models: rows repacked as planes for the matrix-core launch)
    for layers, strat, n_mlp, cfgn in ((5, "S20", 5, "LLAMA3_8B"), (8, "S22", 3, "LLAMA3_8B"), (4, "S40", 3, "LLAMA3_8B"), (4, "S40", 3, "LLAMA32_3B"),
                                       (3, "S20", 3, "LLAMA32_3B"), (4, "S30", 3, "LLAMA32_3B"), (4, "S30", 3, "LLAMA3_8B")):
        new, old = _hb({"GBX_DATTN": "1", "GBX_DMLP": "1"}, layers, 12, strat, cfgn), _hb({"GBX_DATTN": "0"}, layers, 12, strat, cfgn)
        assert int(new["DATTN"]) == layers and int(new["DMLP"]) == n_mlp, (new["DATTN"], new["DMLP"])
        a, b = f(new, "HB"), f(old, "HB")
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= 0.06 * np.abs(b).mean() + 0.06 * np.abs(b).max(), f"{strat}: residual stream after 12 steps"
        ka, kb = f(new, "KC"), f(old, "KC")
        assert np.abs(ka - kb).max() <= 0.06 * np.abs(kb).mean() + 0.06 * np.abs(kb).max(), f"{strat}: last layer's cached keys"


def test_repack_planes_is_the_oracle_unpack_rearranged():
    """gbx_repack_planes (3-bit rows -> a 2-bit and a 1-bit plane per 32 values, what the matrix-core fused MLP launch reads) against the oracle's
    bit-stream unpack (oracle/gba_oracle.py::unpack): words 0 / 1 of a block are the low two bits of its values as 2-bit
    fields in value order, word 2 carries the high bits at 4 n + (0, 1, 2, 3) for values 2 n, 2 n + 1, 16 + 2 n, 17 + 2 n.  Integer work: bit-exact."""
    from types import SimpleNamespace
    sys.path.insert(0, ROOT)
    from oracle import gba_oracle as O
    from gbx_lm_amd import ops
    rng = np.random.default_rng(3)
    for n, k in ((5, 32), (64, 4096), (7, 3072)):
        qw = rng.integers(0, 2**32, size=(n, k * 3 // 32), dtype=np.uint64).astype(np.uint32)
        q = O.unpack(qw, 3).reshape(n, k // 32, 32)                         # [row, block, value]
        lo, hi = (q & 3).astype(np.uint64), (q >> 2).astype(np.uint64)
        w0 = sum(lo[:, :, j] << np.uint64(2 * j) for j in range(16))
        w1 = sum(lo[:, :, 16 + j] << np.uint64(2 * j) for j in range(16))
        w2 = sum((hi[:, :, 2 * m] << np.uint64(4 * m)) | (hi[:, :, 2 * m + 1] << np.uint64(4 * m + 1)) |
                 (hi[:, :, 16 + 2 * m] << np.uint64(4 * m + 2)) | (hi[:, :, 17 + 2 * m] << np.uint64(4 * m + 3)) for m in range(8))
        want = np.stack([w0, w1, w2], axis=-1).astype(np.uint32).reshape(n, -1)
        mod = SimpleNamespace(qweight=torch.from_numpy(qw.view(np.int32)).view(torch.uint32).to(DEV), bits=3, input_dims=k,
                              scales=torch.zeros(1, device=DEV), zeros=torch.zeros(1, device=DEV))
        got = ops.repack_planes(mod).cpu().view(torch.int32).numpy().view(np.uint32)
        assert np.array_equal(got, want), (n, k)
        assert ops.repack_planes(mod).data_ptr() == ops.repack_planes(mod).data_ptr(), "built once per module"
