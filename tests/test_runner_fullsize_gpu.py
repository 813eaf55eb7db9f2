"""GPU: the launches the bench times, as the decode runner assembles them -- ONE decoder layer of Llama-3-8B shape per bit width
through FusedDecodeRunner (graph replay, decode-block GEMV launches with the carried RMSNorm statistic) against the same layer
computed from gbx_dequantize weights in fp64 with the reference's op-by-op roundings (gbx_lm/models/qllama.py:68-141), and
the A/B switches of the launch structure against each other on the same weights."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_layer_model(bits, vocab=1024):
    from gbx_lm_amd import synthetic as S
    cfg = dict(S.LLAMA3_8B, num_hidden_layers=1, vocab_size=vocab)
    strat = S.make_strategy("S22", 1, 128)
    for p in strat["model.layers.0"]:
        strat["model.layers.0"][p] = {"bits": [bits], "group_size": {str(bits): 128}}
    return S.build_model(cfg, strat, device=DEV, seed=bits), cfg


def _bf(x):
    return x.to(torch.bfloat16).double()


def _ref_layer_fp64(model, h, pos, kcache, vcache):
    """One decode step of the layer on fp64 x . dequantised weights, every intermediate rounded to bf16 where the reference graph
    holds a bf16 tensor.  h: [D] bf16 stream, caches [n_kv, T, hd] bf16 (positions < pos valid).  Returns the new stream."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import apply_rope
    blk = model.model.layers[0]
    at, mlp = blk.self_attn, blk.mlp
    deq = lambda m: ops.dequantize(m.qweight, m.scales, m.zeros, m.group_size, m.bits, dtype=torch.float32).double()

    def rms(x, w, eps):
        xf = x.double()
        return _bf(xf * torch.rsqrt(xf.pow(2).mean() + eps) * w.double())

    hd, nh, nkv = at.head_dim, at.n_heads, at.n_kv_heads
    xn = rms(h, blk.input_layernorm.weight, blk.input_layernorm.eps)
    q, k, v = _bf(deq(at.q_proj) @ xn), _bf(deq(at.k_proj) @ xn), _bf(deq(at.v_proj) @ xn)
    inv = at.rope.inv_freq.cpu()
    qr = apply_rope(q.to(torch.bfloat16).cpu().view(1, nh, 1, hd), inv, pos, getattr(at.rope, "scale", 1.0), at.rope.traditional).to(DEV).double().view(nh, hd)
    kr = apply_rope(k.to(torch.bfloat16).cpu().view(1, nkv, 1, hd), inv, pos, getattr(at.rope, "scale", 1.0), at.rope.traditional).to(DEV).double().view(nkv, hd)
    K = torch.cat([kcache[:, :pos].double(), kr[:, None]], 1)            # [nkv, pos + 1, hd]
    V = torch.cat([vcache[:, :pos].double(), v.view(nkv, 1, hd)], 1)
    rep = nh // nkv
    s = torch.einsum("hd,htd->ht", qr, K.repeat_interleave(rep, 0)) * at.scale
    o = _bf(torch.einsum("ht,htd->hd", torch.softmax(s, -1), V.repeat_interleave(rep, 0)).reshape(-1))
    h1 = _bf(h.double() + _bf(deq(at.o_proj) @ o))
    xn2 = rms(h1.to(torch.bfloat16), blk.post_attention_layernorm.weight, blk.post_attention_layernorm.eps)
    g, u = _bf(deq(mlp.gate_proj) @ xn2), _bf(deq(mlp.up_proj) @ xn2)
    act = _bf(_bf(g / (1 + torch.exp(-g))) * u)
    return _bf(h1 + _bf(deq(mlp.down_proj) @ act))


@pytest.mark.parametrize("dattn", [0, 1, 2, 3])
@pytest.mark.parametrize("bits", [2, 3, 4])
def test_one_8b_layer_through_the_fused_runner_vs_fp64(bits, dattn, monkeypatch):
    import bench
    from gbx_lm_amd import ops
    from gbx_lm_amd.decode import FusedDecodeRunner
    # 0: five launches per block (round 4), 1: three, 2: two (round 5: 2-bit blocks; round 6: 3- and 4-bit blocks too), 3: ONE (round 6, 2-bit blocks)
    monkeypatch.setenv("GBX_DATTN", str(min(dattn, 1)))
    monkeypatch.setenv("GBX_DMLP", "1" if dattn >= 2 else "0")
    monkeypatch.setenv("GBX_DBLOCK", "1" if dattn == 3 else "0")
    monkeypatch.setenv("GBX_DMLP_VEC", "1")                   # (the 3- / 4-bit fused MLP launch is opt-in: measured slower than its two launches)
    model, cfg = _one_layer_model(bits)
    ctx = 9
    caches = bench.make_caches(model, 1, 256, DEV)
    runner = FusedDecodeRunner(model, caches, use_graph=True)
    assert (runner.da[0] is not None) == bool(dattn), "the launch path under test"
    assert (runner.db[0] is not None) == (dattn == 3 and bits == 2), "the one-launch block covers the 2-bit block only"
    if dattn >= 2:
        d_a, d_m = runner.da[0]
        assert d_m.x_acc == d_a.acc and runner.da_post is not None, "attention + MLP as two launches, the sums finished after the last block"
        assert (d_m.gate_up_bits, d_m.down_bits) == (bits, bits)
    elif dattn:
        d_a, d_gu3, d_dn3 = runner.da[0]
        assert d_gu3.x_acc == d_a.acc and d_gu3.x_out == d_dn3.residual
    # the launches under test are the decode-block GEMV ones (the bench's), all four, with the statistic carried into gate|up
    g_qkv, g_o, g_gu, g_dn = runner.dg[0]
    assert all(g is not None for g in (g_qkv, g_o, g_gu, g_dn)), "the 8B launches must run on gbx_dgemv_forward"
    assert g_qkv.norm_weight and g_gu.norm_weight and g_gu.epilogue == 2 and g_o.epilogue == 1 and g_dn.epilogue == 1
    if bits == 2:
        assert g_o.ssq_out and g_gu.ssq_in == g_o.ssq_out and g_dn.ssq_out, "2-bit layers carry the RMSNorm statistic"
        assert (g_qkv.rows_per_batch, g_o.rows_per_batch, g_gu.rows_per_batch, g_dn.rows_per_batch) == (2, 4, 2, 2)
    for t in range(ctx):                                    # fill the cache through the decode path
        runner.step(torch.tensor([t + 3], device=DEV))
    tok = 77
    c = caches[0]
    kc, vc = c.keys[0, :, :ctx].clone(), c.values[0, :, :ctx].clone()
    runner.step(torch.tensor([tok], device=DEV))
    h_got = runner.hb.clone()                               # the residual stream after the layer (before the final norm)
    runner.dattn_check()
    h0 = model.model.embed_tokens.weight[tok]
    want = _ref_layer_fp64(model, h0, ctx, kc, vc)
    err = (h_got.double() - want).abs()
    tol = 2e-2 * want.abs() + 2e-2 * want.pow(2).mean().sqrt()
    assert (err <= tol).all(), f"bits {bits}: {int((err > tol).sum())} of {want.numel()} outside tolerance, worst {err.max().item():.4g}"
    # the cache row the step appended is the reference's RoPE(k), v
    assert torch.isfinite(c.keys[0, :, ctx].float()).all() and torch.isfinite(c.values[0, :, ctx].float()).all()


@pytest.mark.parametrize("bits", [2, 4])
def test_long_cache_keeps_the_fused_launches(bits, monkeypatch):
    """A cache of 2048 positions (the five-launch block would split the context over workgroups there): the fused attention launch
    stays, with the kv group's producers sweeping the context in chunks; 1100 cached positions, against the fp64 layer."""
    import bench
    from gbx_lm_amd.decode import FusedDecodeRunner
    monkeypatch.setenv("GBX_DATTN", "1")
    monkeypatch.setenv("GBX_DMLP", "1")
    model, cfg = _one_layer_model(bits)
    ctx = 1100
    caches = bench.make_caches(model, 1, 2048, DEV)
    runner = FusedDecodeRunner(model, caches, use_graph=True)
    assert runner.attn_ws is not None and runner.da[0] is not None, "long cache, fused launches"
    for t in range(ctx):
        runner.step(torch.tensor([(7 * t + 3) % 256], device=DEV))
    tok = 77
    c = caches[0]
    kc, vc = c.keys[0, :, :ctx].clone(), c.values[0, :, :ctx].clone()
    runner.step(torch.tensor([tok], device=DEV))
    h_got = runner.hb.clone()
    runner.dattn_check()
    want = _ref_layer_fp64(model, model.model.embed_tokens.weight[tok], ctx, kc, vc)
    err = (h_got.double() - want).abs()
    tol = 2e-2 * want.abs() + 2e-2 * want.pow(2).mean().sqrt()
    assert (err <= tol).all(), f"bits {bits}: {int((err > tol).sum())} of {want.numel()} outside tolerance, worst {err.max().item():.4g}"


def _tokens(env_extra):
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "import bench\n"
        "from gbx_lm_amd import synthetic as S\n"
        "from gbx_lm_amd.decode import FusedDecodeRunner\n"
        "cfg = dict(S.LLAMA3_8B, num_hidden_layers=3, vocab_size=2048)\n"
        "model = S.build_model(cfg, S.make_strategy('S22', 3, 128), device='cuda:0', seed=5)\n"
        "caches = bench.make_caches(model, 1, 256, 'cuda:0')\n"
        "r = FusedDecodeRunner(model, caches, use_graph=True)\n"
        "for t in range(12): lg = r.step(torch.tensor([(7 * t + 1) %% 2048], device='cuda:0'))\n"     # teacher-forced: no argmax ties
        "torch.cuda.synchronize()\n"
        "print('LOGITS', r.logits_buf.float().cpu().numpy().tobytes().hex()[:4096])\n"
        "print('HB', r.hb.float().cpu().numpy().tobytes().hex())\n" % ROOT)
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = {l.split()[0]: l.split()[1] for l in out.stdout.splitlines() if l.startswith(("LOGITS", "HB"))}
    return lines


def test_launch_structure_switches_agree_on_8b_shapes():
    """Three 8B-shaped layers (S22 on three layers: 4, 3 and 4 bits... whatever the strategy assigns), 12 teacher-forced steps: the
    decode-block GEMV path with and without the carried statistic, and the previous kernel (GBX_DGEMV=0) with q|k|v concatenated
    or not.  Switches that do not change the arithmetic order give identical bits; the carried statistic and a different
    workgroup width may move rstd by an ulp."""
    base = _tokens({})
    assert _tokens({"GBX_DG_SSQ": "0"})["HB"] is not None
    old = _tokens({"GBX_DGEMV": "0"})
    old_nocat = _tokens({"GBX_DGEMV": "0", "GBX_QKV_CAT": "0"})
    assert old["HB"] == old_nocat["HB"], "GBX_QKV_CAT only changes the launch structure"
    new_nossq = _tokens({"GBX_DG_SSQ": "0"})
    late0 = _tokens({"GBX_DG_SSQ": "0", "GBX_DG_LATE": "0"})
    assert new_nossq["HB"] == late0["HB"], "GBX_DG_LATE only changes when the weights are requested"

    def stream(d):
        return np.frombuffer(bytes.fromhex(d["HB"]), dtype=np.float32)
    a, b, c = stream(base), stream(new_nossq), stream(old)
    scale = np.abs(c).mean() + 1e-9
    assert np.abs(b - c).max() <= 0.05 * scale + 0.05 * np.abs(c).max(), "dgemv vs gemv_kernel after 12 steps"
    assert np.abs(a - b).max() <= 0.05 * scale + 0.05 * np.abs(b).max(), "carried statistic vs in-kernel statistic"
