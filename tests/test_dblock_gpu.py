"""GPU: the one-launch decoder block of round 6 (gbx_dblock_forward: input norm -> q|k|v -> RoPE + cache append -> attention -> o_proj
-> chip-wide release -> x' = h + o, post-attention norm -> gate|up from LDS + SwiGLU -> XCD-local hand-off -> down_proj), through the C-ABI.

Reference: the same block in fp64 on gbx_dequantize weights with the reference graph's roundings (gbx_lm/models/qllama.py:68-96,114-115,
131-141; mx.quantized_matmul at M == 1, quantized_linear_gba.py:195-203), tolerance of the matrix-core form (2e-2 of the value + 2e-2 of
the vector's RMS, stated at each assert).  Exact checks: x' = round(h + round(o_proj sums * 2^-32)) bit for bit from the launch's own
accumulator; the appended cache row's neighbours; the clears; bit-reproducibility over repeated launches (fixed-point partial sums are
order-independent); and the attention half bit-identical to gbx_dattn_forward (same arithmetic, the LDS-DMA stream beside it must not
change a bit).
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_dattn_gpu import DEV, _bf, _cat_qkv, _close, _deq, _layer, _reference  # noqa: E402

pytestmark = pytest.mark.gpu


def _descs(model, cfg, ctx, seed, x_acc=False):
    from gbx_lm_amd import ops
    blk = model.model.layers[0]
    at, mlp = blk.self_attn, blk.mlp
    D, I, nh, nkv, hd = cfg["hidden_size"], cfg["intermediate_size"], at.n_heads, at.n_kv_heads, at.head_dim
    cat = _cat_qkv(at)
    cap = 256
    while cap <= ctx:
        cap *= 2
    gen = torch.Generator(device="cpu").manual_seed(seed)
    t = {}
    t["h"] = (torch.randn(D, generator=gen) * 0.7).to(torch.bfloat16).to(DEV)
    t["kc"] = (torch.randn(1, nkv, cap, hd, generator=gen) * 0.5).to(torch.bfloat16).to(DEV)
    t["vc"] = torch.randn(1, nkv, cap, hd, generator=gen).to(torch.bfloat16).to(DEV)
    t["kc0"], t["vc0"] = t["kc"].clone(), t["vc"].clone()
    t["pos"] = torch.tensor([ctx], dtype=torch.long, device=DEV)
    t["acc_o"] = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
    t["acc_d"] = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
    t["attn_out"] = torch.zeros(nh * hd, dtype=torch.bfloat16, device=DEV)
    t["act"] = torch.zeros(I, dtype=torch.bfloat16, device=DEV)
    t["x2"] = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
    t["junk0"] = torch.full((D + 1,), 5, dtype=torch.int64, device=DEV)
    t["junk1"] = torch.full((D + 1,), 7, dtype=torch.int64, device=DEV)
    t["st_a"], t["st_m"], t["st_b"] = ops.dattn_state(DEV), ops.dattn_state(DEV), ops.dblock_state(DEV)
    kw = {}
    if x_acc:      # the stream enters as (vector, fixed-point sums of the previous block's down_proj)
        t["prev"] = torch.zeros(D + 1, dtype=torch.int64, device=DEV)
        t["prev"][:D] = torch.round((torch.randn(D, generator=gen) * 0.2).double() * 2.0 ** 32).to(torch.int64).to(DEV)
        t["hn"] = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
        kw = dict(x_acc=t["prev"], x_out=t["hn"])
    t["pk_qkv"], t["pk_gu"], t["pk_dn"] = ops.dgemv_pack(cat, 2), ops.dgemv_pack(mlp.gate_proj, 2, mlp.up_proj), ops.dgemv_pack(mlp.down_proj, 2)
    d_a = ops.make_dattn_desc(t["h"], blk.input_layernorm.weight, blk.input_layernorm.eps, cat, t["pk_qkv"], t["kc"], t["vc"], at.rope.inv_freq,
                              t["pos"], at.o_proj, t["acc_o"], t["st_a"], nh, cap, getattr(at.rope, "scale", 1.0), at.scale, at.rope.traditional,
                              attn_out=t["attn_out"], zero_buf=t["junk0"], **kw)
    h_buf = t["hn"] if x_acc else t["h"]
    d_m = ops.make_dmlp_desc(h_buf, t["acc_o"], t["x2"], blk.post_attention_layernorm.weight, blk.post_attention_layernorm.eps, mlp.gate_proj,
                             mlp.up_proj, mlp.down_proj, t["pk_gu"], t["pk_dn"], t["acc_d"], t["st_m"], zero_buf=t["junk1"], act_out=t["act"])
    d_b = ops.make_dblock_desc(d_a, d_m, t["st_b"])
    t["cat"] = cat
    return blk, d_a, d_m, d_b, t


@pytest.mark.parametrize("ctx,x_acc", [(0, False), (9, False), (40, True), (128, False), (161, True), (200, False), (330, True), (384, False)])
def test_dblock_launch_vs_fp64(ctx, x_acc):
    from gbx_lm_amd import ops
    model, cfg = _layer("LLAMA3_8B", 2, 2, seed=60 + ctx % 7)
    blk, d_a, d_m, d_b, t = _descs(model, cfg, ctx, seed=1000 + ctx, x_acc=x_acc)
    at, mlp = blk.self_attn, blk.mlp
    D = cfg["hidden_size"]
    assert ops.dblock_supported(D, at.n_heads, at.n_kv_heads, at.head_dim, t["cat"], at.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj, torch.bfloat16)
    assert ctx <= ops.dblock_max_context()
    ops.dblock(d_b, DEV)
    torch.cuda.synchronize()
    assert ops.dblock_status(t["st_b"]) == 0, "a hand-off timed out"
    h = t["h"]
    if x_acc:
        h = (t["h"].double() + (t["prev"][:D].double() * 2.0 ** -32).to(torch.bfloat16).double()).to(torch.bfloat16)
        assert torch.equal(t["hn"], h), "x_out is not round(x + round(x_acc * 2^-32))"
    # ---- attention half (as tests/test_dattn_gpu.py)
    kr, v, attn = _reference(blk, h, ctx, t["kc0"][0], t["vc0"][0])
    _close(t["kc"][0, :, ctx], kr, "appended K row", rel=1e-2)
    _close(t["vc"][0, :, ctx], v, "appended V row", rel=2e-2)
    keep = torch.ones(t["kc"].shape[2], dtype=torch.bool, device=DEV)
    keep[ctx] = False
    assert torch.equal(t["kc"][0][:, keep], t["kc0"][0][:, keep]) and torch.equal(t["vc"][0][:, keep], t["vc0"][0][:, keep]), "rows other than `pos` changed"
    _close(t["attn_out"], attn, "attention output")
    o_got = t["acc_o"][:D].double() * 2.0 ** -32
    _close(o_got, _deq(at.o_proj) @ t["attn_out"].double(), "o_proj partial sums", rel=1e-2)
    assert int(t["acc_o"][D].item()) == 0 and int(t["acc_d"][D].item()) == 0, "non-finite marker set on finite data"
    # ---- the release: x' from the launch's own sums, exact
    want_x = (h.double() + (t["acc_o"][:D].double() * 2.0 ** -32).to(torch.bfloat16).double()).to(torch.bfloat16)
    assert torch.equal(t["x2"], want_x), "x2_out is not round(h + round(acc_o * 2^-32))"
    assert int(t["junk0"].abs().sum().item()) == 0 and int(t["junk1"].abs().sum().item()) == 0, "a zero_buf left words behind"
    # ---- MLP half (as test_dmlp_launch_vs_fp64)
    nw = blk.post_attention_layernorm
    xf = want_x.double()
    xn = _bf(xf * torch.rsqrt(xf.pow(2).mean() + nw.eps) * nw.weight.double())
    g, u = _bf(_deq(mlp.gate_proj) @ xn), _bf(_deq(mlp.up_proj) @ xn)
    _close(t["act"], _bf(_bf(g / (1 + torch.exp(-g))) * u), "SwiGLU activations")
    _close(t["acc_d"][:D].double() * 2.0 ** -32, _deq(mlp.down_proj) @ t["act"].double(), "down_proj sums")


def test_dblock_matches_the_two_launches_and_is_bit_reproducible():
    """Same inputs through gbx_dattn_forward + gbx_dmlp_forward and through gbx_dblock_forward: the attention half (cache row, attention
    output, o_proj's fixed-point sums, x') is bit-identical -- the weight stream beside the chain must not change a bit -- the MLP half
    agrees to the last place of the statistic (the leader sums x'^2 over 16 waves, the MLP launch over 8); 30 launches on one state buffer
    give identical bits (generation tags, order-independent sums)."""
    from gbx_lm_amd import ops
    model, cfg = _layer("LLAMA3_8B", 2, 2, seed=71)
    D = cfg["hidden_size"]
    blk, d_a, d_m, d_b, t = _descs(model, cfg, 77, seed=5, x_acc=True)
    ops.dattn(d_a, DEV)
    ops.dmlp(d_m, DEV)
    torch.cuda.synchronize()
    assert ops.dattn_status(t["st_a"]) == 0 and ops.dattn_status(t["st_m"]) == 0
    two = {k: t[k].clone() for k in ("kc", "vc", "attn_out", "acc_o", "acc_d", "x2", "act", "hn")}
    first = None
    for rep in range(30):
        for k in ("acc_o", "acc_d", "x2", "act", "attn_out", "hn"):
            t[k].zero_()
        t["kc"].copy_(t["kc0"]); t["vc"].copy_(t["vc0"])
        ops.dblock(d_b, DEV)
        torch.cuda.synchronize()
        assert ops.dblock_status(t["st_b"]) == 0
        got = {k: t[k].clone() for k in two}
        if first is None:
            first = got
            for k in ("kc", "vc", "attn_out", "acc_o", "x2", "hn"):
                assert torch.equal(got[k], two[k]), f"{k}: one launch vs two launches differ"
            _close(got["act"], two["act"].double(), "activations vs the two-launch block", rel=1e-2)
            _close(got["acc_d"][:D].double(), two["acc_d"][:D].double(), "down_proj sums vs the two-launch block", rel=2e-2)
        else:
            for k in got:
                assert torch.equal(got[k], first[k]), f"launch {rep}: {k} differs from the first launch"


def test_dblock_non_finite_and_zero_streams():
    """All-zero stream -> exactly zero sums; a NaN in the stream reaches both accumulators' markers and the next consumer produces NaN."""
    from gbx_lm_amd import ops
    model, cfg = _layer("LLAMA3_8B", 2, 2, seed=72)
    D = cfg["hidden_size"]
    for bad in (None, float("nan")):
        blk, d_a, d_m, d_b, t = _descs(model, cfg, 33, seed=9)
        t["h"].zero_()
        t["kc"].zero_(); t["vc"].zero_()
        if bad is not None:
            t["h"].copy_(torch.randn(D).to(torch.bfloat16))
            t["h"][D // 2 + 3] = bad
        ops.dblock(d_b, DEV)
        torch.cuda.synchronize()
        assert ops.dblock_status(t["st_b"]) == 0
        if bad is None:
            assert int(t["acc_o"].abs().sum().item()) == 0 and int(t["acc_d"].abs().sum().item()) == 0, "zero stream must give exactly zero sums"
        else:
            assert int(t["acc_d"][D].item()) > 0, "non-finite marker of the down_proj accumulator not raised"
            out = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
            ops.acc_residual(t["x2"], t["acc_d"], out)
            assert torch.isnan(out.float()).all()


def test_runner_switches_graphs_at_the_context_limit(monkeypatch):
    """Three 2-bit 8B layers through FusedDecodeRunner across gbx_dblock_max_context(): the one-launch graph up to the limit, the two-launch
    graph (cooperative sweep) beyond, against the same teacher-forced steps with GBX_DBLOCK=0 -- streams agree to a few bf16 ulps of the
    stream's scale at every step (different summation order of one statistic), the appended cache rows likewise."""
    import bench
    from gbx_lm_amd import ops, synthetic as S
    from gbx_lm_amd.decode import FusedDecodeRunner
    cfg = dict(S.LLAMA3_8B, num_hidden_layers=3, vocab_size=2048)
    strat = S.make_strategy("S22", 3, 128)
    for li in range(3):
        for p in strat[f"model.layers.{li}"]:
            strat[f"model.layers.{li}"][p] = {"bits": [2], "group_size": {"2": 128}}
    model = S.build_model(cfg, strat, device=DEV, seed=4)
    limit = ops.dblock_max_context()

    def run(dblock):
        monkeypatch.setenv("GBX_DBLOCK", dblock)
        caches = bench.make_caches(model, 1, 1024, DEV)
        for c in caches:                              # a context just below the limit, as a prompt pass would have left it
            g = torch.Generator(device="cpu").manual_seed(3)
            c.keys[:, :, :limit - 4] = (torch.randn(1, 8, limit - 4, 128, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
            c.values[:, :, :limit - 4] = torch.randn(1, 8, limit - 4, 128, generator=g).to(torch.bfloat16).to(DEV)
            c.offset = limit - 4
        r = FusedDecodeRunner(model, caches, use_graph=True)
        assert all((d is not None) == (dblock == "1") for d in r.db)
        outs = []
        for t in range(10):
            r.step(torch.tensor([(11 * t + 5) % 2048], device=DEV))
            outs.append(r.hb.clone())
        torch.cuda.synchronize()
        r.dattn_check()
        modes = sorted(r._graphs)
        return outs, modes, [c.keys[0, :, limit - 4:limit + 6].clone() for c in caches]

    a, modes_a, ka = run("1")
    b, modes_b, kb = run("0")
    assert modes_a == [(False, False), (False, True)], "both graphs were captured and used"
    assert modes_b == [(False, False)]
    for t, (u, v) in enumerate(zip(a, b)):
        scale = v.float().abs().mean().item()
        assert (u.float() - v.float()).abs().max().item() <= 0.06 * scale + 0.03 * v.float().abs().max().item(), f"step {t}"
    for u, v in zip(ka, kb):
        assert (u.float() - v.float()).abs().max().item() <= 0.05 * v.float().abs().max().item()
