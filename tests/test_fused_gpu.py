"""GPU: the fused decode-block entry points (C-ABI) against the oracle / a plain PyTorch fp32 restatement of the same
ops, then the whole fused graph-replayed decode against the unfused host graph and the CPU-oracle model."""
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


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("k", [256, 4096])
def test_fused_norm_qkv(oracle, bits, k):
    from gbx_lm_amd import ops
    g = 128 if bits != 3 else 32
    mods, ds = zip(*[_mk(oracle, n, k, bits, g, seed=10 * bits + i) for i, n in enumerate((512, 128, 130))])
    x = (torch.randn(k) * 3).to(torch.bfloat16).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(torch.bfloat16).to(DEV)
    ys = [torch.zeros(m.output_dims, dtype=torch.bfloat16, device=DEV) for m in mods]
    desc = ops.make_fused_desc(x, mods, ys, norm_weight=nw, eps=1e-5)
    ops.qlinear_fused(desc, x.device)
    xn = _rms(x, nw, 1e-5).cpu()                              # bf16-rounded norm output, like the reference graph
    for m, d, y in zip(mods, ds, ys):
        y64 = oracle.qlinear_ref(xn[None], d["qweight"], d["scales"], d["zeros"], bits, g, acc=np.float64)
        oracle.assert_close(y[None], y64, "bf16", f"fused qkv b{bits} K{k}")


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("n,k", [(256, 512), (4096, 14336), (130, 4096)])
def test_fused_residual(oracle, bits, n, k):
    from gbx_lm_amd import ops
    m, d = _mk(oracle, n, k, bits, 128, seed=bits + n)
    x = d["x"].to(DEV).reshape(-1)
    h = torch.randn(n).to(torch.bfloat16).to(DEV)
    h0 = h.clone()
    desc = ops.make_fused_desc(x, (m,), (h,), residual=h)     # in place: h <- h + W x
    ops.qlinear_fused(desc, x.device)
    y64 = oracle.qlinear_ref(d["x"], d["qweight"], d["scales"], d["zeros"], bits, 128, acc=np.float64)
    want = h0.cpu().double().numpy()[None] + y64
    oracle.assert_close(h[None], want, "bf16", "residual")


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_fused_norm_swiglu(oracle, bits):
    from gbx_lm_amd import ops
    k, n, g = 4096, 1030, 64
    gate, dg = _mk(oracle, n, k, bits, g, seed=1)
    up, du = _mk(oracle, n, k, bits, g, seed=2)
    x = (torch.randn(k) * 2).to(torch.bfloat16).to(DEV)
    nw = torch.ones(k, dtype=torch.bfloat16, device=DEV)
    act = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(x, (gate, up), (act,), norm_weight=nw, eps=1e-5, swiglu=True), x.device)
    xn = _rms(x, nw, 1e-5).cpu()
    g64 = oracle.qlinear_ref(xn[None], dg["qweight"], dg["scales"], dg["zeros"], bits, g, acc=np.float64)
    u64 = oracle.qlinear_ref(xn[None], du["qweight"], du["scales"], du["zeros"], bits, g, acc=np.float64)
    want = (g64 / (1 + np.exp(-g64))) * u64
    oracle.assert_close(act[None], want, "bf16", "swiglu")


@pytest.mark.parametrize("hd,nh,nkv", [(128, 8, 2), (64, 4, 2), (128, 4, 4)])
@pytest.mark.parametrize("pos", [0, 1, 5, 63, 200])
def test_attn_decode_matches_torch(hd, nh, nkv, pos):
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import RoPE, apply_rope
    torch.manual_seed(pos + hd)
    cap = 256
    q = torch.randn(nh * hd).to(torch.bfloat16).to(DEV)
    k = torch.randn(nkv * hd).to(torch.bfloat16).to(DEV)
    v = torch.randn(nkv * hd).to(torch.bfloat16).to(DEV)
    kc = torch.randn(1, nkv, cap, hd).to(torch.bfloat16).to(DEV)
    vc = torch.randn(1, nkv, cap, hd).to(torch.bfloat16).to(DEV)
    rope = RoPE(hd, base=500000.0).to(DEV)
    kc0, vc0 = kc.clone(), vc.clone()
    out = torch.zeros(nh * hd, dtype=torch.bfloat16, device=DEV)
    p = torch.tensor([pos], dtype=torch.long, device=DEV)
    ops.attn_decode(q, k, v, kc, vc, rope.inv_freq, p, out, nh, nkv, hd, cap, 1.0, hd ** -0.5, False)
    # reference: torch fp32
    inv = rope.inv_freq.cpu()                       # reference RoPE: the torch fp32 formulation on the CPU
    qr = apply_rope(q.cpu().view(1, nh, 1, hd), inv, pos, 1.0, False).to(DEV)
    kr = apply_rope(k.cpu().view(1, nkv, 1, hd), inv, pos, 1.0, False).to(DEV)
    kc0[:, :, pos:pos + 1] = kr
    vc0[:, :, pos:pos + 1] = v.view(1, nkv, 1, hd)
    assert torch.equal(kc, kc0) or torch.allclose(kc.float(), kc0.float(), atol=2e-2), "cache append (k)"
    assert torch.equal(vc, vc0), "cache append (v)"
    rep = nh // nkv
    K = kc0[:, :, :pos + 1].repeat_interleave(rep, 1).float()
    V = vc0[:, :, :pos + 1].repeat_interleave(rep, 1).float()
    s = (qr.float() @ K.transpose(-1, -2)) * hd ** -0.5
    want = (torch.softmax(s, -1) @ V).reshape(-1)
    assert torch.allclose(out.float(), want, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("hd,nh,nkv,slices,B,normed", [(128, 32, 8, 4, 32, False), (128, 8, 2, 7, 3, True), (64, 4, 4, 1, 2, False), (128, 6, 2, 8, 5, False)])
def test_attn_decode_slices_is_the_summed_projection_bit_for_bit(hd, nh, nkv, slices, B, normed):
    """gbx_attn_decode_slices (q | k | v as the fp32 slices of one k-split GEMM) against gbx_attn_decode_ws on the slices summed and
    rounded the way the GEMM's own summing launch does (v = ws[0]; v += ws[s]; one rounding): outputs and both caches identical."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import RoPE
    torch.manual_seed(hd + slices)
    cap, pos = 192, 137
    n = (nh + 2 * nkv) * hd
    ws = (torch.randn(slices, B, n, device=DEV) * 0.6).float()
    acc = ws[0].clone()
    for sl in range(1, slices):
        acc += ws[sl]
    qkv = acc.to(torch.bfloat16)
    q, k, v = qkv[:, :nh * hd].contiguous(), qkv[:, nh * hd:(nh + nkv) * hd].contiguous(), qkv[:, (nh + nkv) * hd:].contiguous()
    kc = torch.randn(B, nkv, cap, hd, device=DEV).to(torch.bfloat16)
    vc = torch.randn(B, nkv, cap, hd, device=DEV).to(torch.bfloat16)
    rope = RoPE(hd, traditional=False, base=500000.0)
    inv = rope.inv_freq.to(DEV)
    qn = (torch.rand(hd, device=DEV) + 0.5).to(torch.bfloat16) if normed else None
    kn = (torch.rand(hd, device=DEV) + 0.5).to(torch.bfloat16) if normed else None
    p = torch.tensor([pos], device=DEV)
    k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    o1 = torch.zeros(B, nh * hd, dtype=torch.bfloat16, device=DEV)
    o2 = torch.zeros_like(o1)
    ops.attn_decode_batched(q, k, v, k1, v1, inv, p, o1, nh, nkv, hd, 1.0, hd ** -0.5, False, qn, kn, 1e-6)
    ops.attn_decode_slices(ws, slices, k2, v2, inv, p, o2, nh, nkv, hd, 1.0, hd ** -0.5, False, qn, kn, 1e-6)
    assert torch.equal(k1, k2) and torch.equal(v1, v2), "cache rows appended from the slices"
    assert torch.equal(o1, o2)


def test_batched_runner_qkv_slices_switch_agrees():
    """DecodeRunner at B = 4 with q|k|v as one k-split GEMM + gbx_attn_decode_slices (the default) against the three-segment launch
    (GBX_QKV_SLICES=0).  The attention launch is bit-identical to the summing launch it replaces (test above); the GEMM itself adds
    its k ranges in a different association when k is split across workgroups, so the logits agree to rounding, not bit for bit."""
    import os
    import bench
    from gbx_lm_amd import synthetic as S
    from gbx_lm_amd.decode import DecodeRunner
    cfg = dict(S.LLAMA3_8B, num_hidden_layers=2, vocab_size=2048)
    model = S.build_model(cfg, S.make_strategy("S22", 2, 128), device=DEV, seed=11)
    outs = []
    for flag in ("1", "0"):
        os.environ["GBX_QKV_SLICES"] = flag
        try:
            r = DecodeRunner(model, bench.make_caches(model, 4, 64, DEV), use_graph=False)
            assert bool(r.qkv_cat) == (flag == "1")
            lg = [r.step(torch.tensor([3 + t, 9, 100 + t, 7], device=DEV)).float().clone() for t in range(5)]
        finally:
            os.environ.pop("GBX_QKV_SLICES", None)
        outs.append(torch.stack(lg))
    assert torch.isfinite(outs[0]).all()
    scale = outs[1].abs().mean().item()
    assert (outs[0] - outs[1]).abs().max().item() <= 0.05 * scale + 0.02 * outs[1].abs().max().item()


@pytest.mark.parametrize("graph", [False, True])
def test_batched_runner_chip_wide_norm_switch_agrees(graph):
    """DecodeRunner at B = 32 with the slice-sum + residual + RMSNorm launches spread over the chip (the default) against the
    row-per-workgroup launch (GBX_NORM_XCD=0), eager and as a replayed graph (the generation words must survive replays): the residual
    stream is bit-identical, the normalised rows agree to the last place, so the logits agree to rounding; the status word stays 0."""
    import os
    import bench
    from gbx_lm_amd import ops, synthetic as S
    from gbx_lm_amd.decode import DecodeRunner
    cfg = dict(S.LLAMA3_8B, num_hidden_layers=2, vocab_size=2048)
    model = S.build_model(cfg, S.make_strategy("S22", 2, 128), device=DEV, seed=12)
    outs = []
    for flag in ("1", "0"):
        os.environ["GBX_NORM_XCD"] = flag
        try:
            r = DecodeRunner(model, bench.make_caches(model, 32, 64, DEV), use_graph=graph)
            lg = [r.step((torch.arange(32, device=DEV) * 7 + 3 + t) % 2048).float().clone() for t in range(12)]
            r.norm_check()
            assert ops.norm_status(r.norm_state) == 0
        finally:
            os.environ.pop("GBX_NORM_XCD", None)
        outs.append(torch.stack(lg))
    assert torch.isfinite(outs[0]).all()
    scale = outs[1].abs().mean().item()
    assert (outs[0] - outs[1]).abs().max().item() <= 0.05 * scale + 0.02 * outs[1].abs().max().item()
    assert (outs[0].argmax(-1) == outs[1].argmax(-1)).float().mean().item() > 0.97


@pytest.mark.parametrize("hd,nh,nkv,traditional", [(128, 8, 2, False), (64, 4, 4, True), (128, 6, 2, True), (128, 4, 2, False)])
def test_attn_decode_batched_matches_torch(hd, nh, nkv, traditional):
    """gbx_attn_decode_batched: B sequences at the same position, each against its own cache rows."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import RoPE, apply_rope
    torch.manual_seed(hd)
    B, cap, pos = 5, 192, 131
    q = torch.randn(B, nh * hd).to(torch.bfloat16)
    k = torch.randn(B, nkv * hd).to(torch.bfloat16)
    v = torch.randn(B, nkv * hd).to(torch.bfloat16)
    kc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16)
    vc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16)
    rope = RoPE(hd, traditional=traditional, base=500000.0)
    kd, vd = kc.to(DEV), vc.to(DEV)
    out = torch.zeros(B, nh * hd, dtype=torch.bfloat16, device=DEV)
    ops.attn_decode_batched(q.to(DEV), k.to(DEV), v.to(DEV), kd, vd, rope.inv_freq.to(DEV), torch.tensor([pos], device=DEV), out,
                            nh, nkv, hd, 1.0, hd ** -0.5, traditional)
    qr = apply_rope(q.view(B, 1, nh, hd).transpose(1, 2), rope.inv_freq, pos, 1.0, traditional)       # [B, nh, 1, hd]
    kr = apply_rope(k.view(B, 1, nkv, hd).transpose(1, 2), rope.inv_freq, pos, 1.0, traditional)
    kc[:, :, pos:pos + 1] = kr
    vc[:, :, pos:pos + 1] = v.view(B, nkv, 1, hd)
    assert torch.allclose(kd.float().cpu(), kc.float(), atol=2e-2) and torch.equal(vd.cpu(), vc)
    rep = nh // nkv
    K = kc[:, :, :pos + 1].repeat_interleave(rep, 1).float()
    V = vc[:, :, :pos + 1].repeat_interleave(rep, 1).float()
    want = (torch.softmax((qr.float() @ K.transpose(-1, -2)) * hd ** -0.5, -1) @ V).reshape(B, -1)
    assert torch.allclose(out.float().cpu(), want, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("B,hd,nh,nkv", [(1, 128, 32, 4), (3, 128, 8, 2), (2, 64, 6, 2), (1, 128, 5, 5)])
def test_attn_decode_with_qk_norm_matches_torch(B, hd, nh, nkv):
    """gbx_attn_decode_normed: per-head RMSNorm of q and k, RoPE, cache append and grouped attention in one launch
    (Qwen3's attention block at decode) against the op-by-op torch formulation; rep = 8 (two workgroups per kv group,
    one of them appends), 4, 3, 1."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import RoPE, apply_rope
    torch.manual_seed(nh + B)
    cap, pos, eps = 96, 41, 1e-6
    q = torch.randn(B, nh * hd).to(torch.bfloat16); k = torch.randn(B, nkv * hd).to(torch.bfloat16)
    v = torch.randn(B, nkv * hd).to(torch.bfloat16)
    kc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16); vc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16)
    qw = (1 + 0.2 * torch.randn(hd)).to(torch.bfloat16); kw = (1 + 0.2 * torch.randn(hd)).to(torch.bfloat16)
    rope = RoPE(hd, base=1000000.0)
    kd, vd = kc.to(DEV), vc.to(DEV)
    out = torch.zeros(B, nh * hd, dtype=torch.bfloat16, device=DEV)
    ops.attn_decode_batched(q.to(DEV), k.to(DEV), v.to(DEV), kd, vd, rope.inv_freq.to(DEV), torch.tensor([pos], device=DEV), out,
                            nh, nkv, hd, 1.0, hd ** -0.5, False, qw.to(DEV), kw.to(DEV), eps)
    def norm(x, w):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()).to(torch.bfloat16)
    qn = norm(q.view(B, 1, nh, hd), qw).transpose(1, 2); kn = norm(k.view(B, 1, nkv, hd), kw).transpose(1, 2)
    qr = apply_rope(qn, rope.inv_freq, pos, 1.0, False); kr = apply_rope(kn, rope.inv_freq, pos, 1.0, False)
    kc[:, :, pos:pos + 1] = kr; vc[:, :, pos:pos + 1] = v.view(B, nkv, 1, hd)
    assert torch.allclose(kd.float().cpu(), kc.float(), atol=3e-2) and torch.equal(vd.cpu(), vc)
    rep = nh // nkv
    K = kc[:, :, :pos + 1].repeat_interleave(rep, 1).float(); V = vc[:, :, :pos + 1].repeat_interleave(rep, 1).float()
    want = (torch.softmax((qr.float() @ K.transpose(-1, -2)) * hd ** -0.5, -1) @ V).reshape(B, -1)
    assert torch.allclose(out.float().cpu(), want, atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("B,nh,nkv,cap,pos", [(1, 32, 8, 4096, 3011), (1, 8, 2, 1536, 5), (2, 8, 2, 8192, 8000), (1, 4, 4, 2048, 1999)])
def test_attn_decode_split_context_matches_torch(B, nh, nkv, cap, pos):
    """Long caches: the grouped kernel with the cached positions split over workgroups + the merge launch
    (gbx_attn_decode_ws) against torch; splits that receive no position (pos = 5 of 1024) must not disturb the merge."""
    from gbx_lm_amd import ops, _lib
    from gbx_lm_amd.models.rope_utils import RoPE, apply_rope
    torch.manual_seed(pos)
    hd = 128
    assert _lib.lib().gbx_attn_workspace_bytes(B, nh, nkv, hd, cap) > 0
    q = torch.randn(B, nh * hd).to(torch.bfloat16); k = torch.randn(B, nkv * hd).to(torch.bfloat16)
    v = torch.randn(B, nkv * hd).to(torch.bfloat16)
    kc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16); vc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16)
    rope = RoPE(hd, base=500000.0)
    kd, vd = kc.to(DEV), vc.to(DEV)
    out = torch.zeros(B, nh * hd, dtype=torch.bfloat16, device=DEV)
    ops.attn_decode_batched(q.to(DEV), k.to(DEV), v.to(DEV), kd, vd, rope.inv_freq.to(DEV), torch.tensor([pos], device=DEV), out,
                            nh, nkv, hd, 1.0, hd ** -0.5, False)
    qr = apply_rope(q.view(B, 1, nh, hd).transpose(1, 2), rope.inv_freq, pos, 1.0, False)
    kr = apply_rope(k.view(B, 1, nkv, hd).transpose(1, 2), rope.inv_freq, pos, 1.0, False)
    kc[:, :, pos:pos + 1] = kr; vc[:, :, pos:pos + 1] = v.view(B, nkv, 1, hd)
    assert torch.allclose(kd.float().cpu(), kc.float(), atol=3e-2) and torch.equal(vd.cpu(), vc)
    rep = nh // nkv
    K = kc[:, :, :pos + 1].repeat_interleave(rep, 1).float(); V = vc[:, :, :pos + 1].repeat_interleave(rep, 1).float()
    want = (torch.softmax((qr.float() @ K.transpose(-1, -2)) * hd ** -0.5, -1) @ V).reshape(B, -1)
    assert torch.allclose(out.float().cpu(), want, atol=2e-2, rtol=2e-2)


def test_attn_decode_with_qkv_bias_matches_torch():
    """Qwen2: the q / k / v projection biases ride in the attention launch (second rounding like the projection's own
    `+ bias`), B = 1 and B = 2."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import RoPE, apply_rope
    torch.manual_seed(4)
    hd, nh, nkv, cap, pos = 128, 4, 2, 64, 19
    for B in (1, 2):
        q = torch.randn(B, nh * hd).to(torch.bfloat16); k = torch.randn(B, nkv * hd).to(torch.bfloat16)
        v = torch.randn(B, nkv * hd).to(torch.bfloat16)
        bq, bk, bv = (torch.randn(n * hd).to(torch.bfloat16) for n in (nh, nkv, nkv))
        kc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16); vc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16)
        rope = RoPE(hd, base=1000000.0)
        kd, vd = kc.to(DEV), vc.to(DEV)
        out = torch.zeros(B, nh * hd, dtype=torch.bfloat16, device=DEV)
        ops.attn_decode_batched(q.to(DEV), k.to(DEV), v.to(DEV), kd, vd, rope.inv_freq.to(DEV), torch.tensor([pos], device=DEV), out,
                                nh, nkv, hd, 1.0, hd ** -0.5, False, None, None, 0.0, bq.to(DEV), bk.to(DEV), bv.to(DEV))
        qb, kb, vb = q + bq, k + bk, v + bv                          # bf16 adds: one rounding
        qr = apply_rope(qb.view(B, 1, nh, hd).transpose(1, 2), rope.inv_freq, pos, 1.0, False)
        kr = apply_rope(kb.view(B, 1, nkv, hd).transpose(1, 2), rope.inv_freq, pos, 1.0, False)
        kc[:, :, pos:pos + 1] = kr; vc[:, :, pos:pos + 1] = vb.view(B, nkv, 1, hd)
        assert torch.allclose(kd.float().cpu(), kc.float(), atol=3e-2) and torch.equal(vd.cpu(), vc)
        rep = nh // nkv
        K = kc[:, :, :pos + 1].repeat_interleave(rep, 1).float(); V = vc[:, :, :pos + 1].repeat_interleave(rep, 1).float()
        want = (torch.softmax((qr.float() @ K.transpose(-1, -2)) * hd ** -0.5, -1) @ V).reshape(B, -1)
        assert torch.allclose(out.float().cpu(), want, atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("hd,nh,nkv", [(128, 8, 2), (128, 32, 4), (64, 6, 2), (64, 5, 5), (128, 4, 2)])
def test_attn_cached_matches_torch(hd, nh, nkv):
    """gbx_attn_cached (what the SDPA helper runs inside a graph-replayed decode step): rotated q against a cache that
    already holds the new token, positions past *pos ignored; rep = 4, 8 (two passes of 4), 3, 1, 2."""
    from gbx_lm_amd import ops
    torch.manual_seed(nh)
    B, cap, pos = 3, 160, 77
    q = torch.randn(B, nh, 1, hd).to(torch.bfloat16)
    kc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16)
    vc = torch.randn(B, nkv, cap, hd).to(torch.bfloat16)
    out = ops.attn_cached(q.to(DEV), kc.to(DEV), vc.to(DEV), torch.tensor([pos], device=DEV), hd ** -0.5)
    rep = nh // nkv
    K = kc[:, :, :pos + 1].repeat_interleave(rep, 1).float()
    V = vc[:, :, :pos + 1].repeat_interleave(rep, 1).float()
    want = torch.softmax((q.float() @ K.transpose(-1, -2)) * hd ** -0.5, -1) @ V
    assert out.shape == q.shape and torch.allclose(out.float().cpu(), want, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_logprobs_argmax_matches_torch(dt):
    """gbx_logprobs_argmax against the two torch ops of generate_step (utils.py:305-306,285), vocabulary sizes that do and
    do not divide by the workgroup count, a tie (lowest index wins) and a dominant token."""
    from gbx_lm_amd import ops
    g = torch.Generator().manual_seed(8)
    for v in (128256, 32000, 517, 100):
        lg = (3 * torch.randn(1, v, generator=g)).to(dt)
        lg[0, v // 3] = lg.max()                                   # a tie with an earlier / later index
        want_lp = lg - torch.logsumexp(lg.float(), -1, keepdim=True).to(dt)
        lp, tok = ops.logprobs_argmax(lg.to(DEV))
        assert lp.shape == lg.shape and tok.dtype == torch.int64 and tok.shape == (1,)
        ulp = 2.0 ** (-7 if dt == torch.bfloat16 else -10)
        assert torch.allclose(lp.float().cpu(), want_lp.float(), rtol=2 * ulp, atol=2 * ulp * want_lp.float().abs().max().item())
        assert int(tok) == int(torch.argmax(lp.cpu(), -1))         # first maximum of what the kernel itself wrote
        assert lp.float().cpu()[0, int(tok)] == lp.float().cpu().max()
    peaked = torch.full((1, 4096), -20.0).to(dt); peaked[0, 77] = 9.0
    lp, tok = ops.logprobs_argmax(peaked.to(DEV))
    assert int(tok) == 77 and abs(float(lp[0, 77])) < 1e-2


def test_kv_append_and_single_token_multi_matrix(oracle):
    """gbx_kv_append writes exactly one position of both caches (different k / v widths); quantized_matmul_multi at
    one token (the multi-segment decode GEMV) equals one call per module bit for bit."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    B, H, cap, dk, dv, pos = 3, 2, 40, 192, 128, 17
    kc = torch.randn(B, H, cap, dk).to(torch.bfloat16).to(DEV); vc = torch.randn(B, H, cap, dv).to(torch.bfloat16).to(DEV)
    k0, v0 = kc.clone(), vc.clone()
    k = torch.randn(B, 1, H, dk).to(torch.bfloat16).to(DEV).permute(0, 2, 1, 3)
    v = torch.randn(B, 1, H, dv).to(torch.bfloat16).to(DEV).permute(0, 2, 1, 3)
    ops.kv_append(kc, vc, k, v, torch.tensor([pos], device=DEV))
    k0[:, :, pos:pos + 1] = k; v0[:, :, pos:pos + 1] = v
    assert torch.equal(kc, k0) and torch.equal(vc, v0)
    mods = []
    for i, n in enumerate((512, 128, 128)):
        d = oracle.synth_layer(n, 1024, 2, 128, seed=40 + i)
        q = QuantizedLinear(1024, n, False, 128, 2)
        q._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
        q._buffers["scales"], q._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
        mods.append(q)
    x = torch.randn(1, 1, 1024).to(torch.bfloat16).to(DEV)
    for y, q in zip(ops.quantized_matmul_multi(x, mods), mods):
        assert y.shape == (1, 1, q.output_dims) and torch.equal(y, q(x))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_add_rmsnorm_and_swiglu_glue(dt):
    """Batched-decode glue kernels against the op-by-op torch formulation (same roundings): h += d; y = RMSNorm(h) w,
    and silu(g) * u."""
    from gbx_lm_amd import ops
    g = torch.Generator().manual_seed(11)
    M, K = 7, 4096
    h = torch.randn(M, 1, K, generator=g).to(dt)
    d = torch.randn(M, 1, K, generator=g).to(dt)
    w = (1 + 0.1 * torch.randn(K, generator=g)).to(dt)
    hd_ = h.to(DEV).clone()
    y = ops.add_rmsnorm(hd_, d.to(DEV), w.to(DEV), 1e-5)
    hs = h + d                                                         # one rounding to dt
    assert torch.equal(hd_.cpu(), hs)
    hf = hs.float()
    want = hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()
    assert torch.allclose(y.float().cpu(), want, atol=2e-2, rtol=1e-2)
    gate = (3 * torch.randn(M, 14336, generator=g)).to(dt)
    up = torch.randn(M, 14336, generator=g).to(dt)
    got = ops.swiglu(gate.to(DEV), up.to(DEV)).cpu()
    ref = (torch.nn.functional.silu(gate.float()).to(dt) * up)          # silu rounded to dt, product rounded to dt
    ulp = 2.0 ** (-7 if dt == torch.bfloat16 else -10)
    assert torch.allclose(got.float(), ref.float(), rtol=2 * ulp, atol=1e-6)


@pytest.mark.parametrize("n,k", [(1000, 256), (128256, 4096), (513, 3072)])
@pytest.mark.parametrize("norm", [False, True])
def test_dense_gemv(n, k, norm):
    from gbx_lm_amd import ops
    torch.manual_seed(n)
    w = (torch.randn(n, k) * 0.02).to(torch.bfloat16).to(DEV)
    x = torch.randn(k).to(torch.bfloat16).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(torch.bfloat16).to(DEV)
    y = ops.dense_gemv(x, w, norm_weight=nw if norm else None, eps=1e-5)
    xin = _rms(x, nw, 1e-5) if norm else x
    want = w.double() @ xin.double()
    rms = want.pow(2).mean().sqrt().item()
    assert torch.allclose(y.double(), want, atol=2e-2 * rms, rtol=2e-2)


def test_rmsnorm_kernel():
    from gbx_lm_amd import ops
    x = torch.randn(5, 3072).to(torch.bfloat16).to(DEV)
    w = (1 + 0.1 * torch.randn(3072)).to(torch.bfloat16).to(DEV)
    y = ops.rmsnorm(x, w, 1e-5)
    assert torch.allclose(y.float(), _rms(x, w, 1e-5).float(), atol=1e-2, rtol=1e-2)


def test_fused_runner_matches_unfused_and_oracle():
    """6-layer mixed 2/3/4-bit tiny Llama: fused graph decode vs unfused graph decode (same GPU) vs the CPU-oracle
    model, teacher-forced."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
    from conftest import oracle_matmul
    from gbx_lm_amd import ops, synthetic as S
    from gbx_lm_amd.decode import DecodeRunner, FusedDecodeRunner, fused_eligible
    from gbx_lm_amd.models.cache import KVCache, make_prompt_cache
    cfg = dict(S.TINY, num_hidden_layers=6, hidden_size=512, intermediate_size=1024, num_attention_heads=4,
               num_key_value_heads=2)      # head_dim 128
    strat = S.make_strategy("S22", 6, 64)
    cpu = S.build_model(cfg, strat, device="cpu", seed=4)
    gpu = S.build_model(cfg, strat, device="cpu", seed=4).to(DEV)
    for blk in gpu.model.layers:
        blk.self_attn.rope.to(DEV)
    assert fused_eligible(gpu)
    prompt = [7, 3, 200, 41]
    ids = [9, 100, 37, 250, 6, 77, 300, 12, 5, 5, 480, 2]
    # CPU oracle model (op monkeypatched for this block only)
    real = ops.quantized_matmul
    try:
        ops.quantized_matmul = oracle_matmul
        c = make_prompt_cache(cpu)
        with torch.no_grad():
            ref = [cpu(torch.tensor([prompt]), cache=c)[:, -1, :].float()]
            for t in ids:
                ref.append(cpu(torch.tensor([[t]]), cache=c)[:, -1, :].float())
    finally:
        ops.quantized_matmul = real
    outs = {}
    for name, cls in (("fused", FusedDecodeRunner), ("unfused", DecodeRunner)):
        caches = [KVCache() for _ in gpu.model.layers]
        with torch.no_grad():
            first = gpu(torch.tensor([prompt], device=DEV), cache=caches)[:, -1, :].float().cpu()
            r = cls(gpu, caches)
            got = [first]
            for t in ids:
                got.append(r.step(torch.tensor([t], device=DEV)).float().cpu().clone())
        outs[name] = got
    for i in range(len(ref)):
        scale = ref[i].abs().max().item()
        for name in outs:
            err = (outs[name][i] - ref[i]).abs().max().item()
            assert err < 0.05 * scale + 0.05, f"{name} step {i}: {err} vs scale {scale}"


def test_fused_argmax_and_device_greedy_chain():
    """lm_head with fused argmax keys + embed_next_token: the on-device greedy chain must produce the same tokens as
    host-driven greedy decoding (argmax by torch, token fed back through step())."""
    from gbx_lm_amd import ops, synthetic as S
    from gbx_lm_amd.decode import FusedDecodeRunner
    from gbx_lm_amd.models.cache import KVCache
    # 1) keys decode to torch.argmax (lowest index on ties)
    torch.manual_seed(0)
    n, k = 5000, 512
    w = (torch.randn(n, k) * 0.05).to(torch.bfloat16).to(DEV)
    w[777] = w[123]                                                   # exact tie between two rows
    x = torch.randn(k).to(torch.bfloat16).to(DEV)
    keys = torch.zeros(64, dtype=torch.int64, device=DEV)
    y = ops.dense_gemv(x, w, argmax_keys=keys)
    h = torch.zeros(k, dtype=torch.bfloat16, device=DEV)
    tok = torch.zeros(1, dtype=torch.long, device=DEV)
    pos = torch.tensor([41], dtype=torch.long, device=DEV)
    table = torch.randn(n, k).to(torch.bfloat16).to(DEV)
    ops.embed_next_token(keys, tok, pos, table, h, from_keys=True, advance_pos=True)
    want = int(torch.argmax(y.float()).item())
    assert int(tok.item()) == want and int(pos.item()) == 42
    assert torch.equal(h, table[want]) and int(keys.abs().sum().item()) == 0
    # 2) whole chain on a small model, graph and eager
    cfg = dict(S.TINY, num_hidden_layers=3, hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=2)
    model = S.build_model(cfg, S.make_strategy("S22", 3, 64), device=DEV, seed=5)
    at = model.model.layers[0].self_attn

    def mk():
        caches = []
        for _ in model.model.layers:
            c = KVCache(step=64)
            like = torch.empty((1, at.n_kv_heads, 1, at.head_dim), dtype=torch.bfloat16, device=DEV)
            c.reserve(1, like, like)
            caches.append(c)
        return caches

    first = torch.tensor([11], device=DEV)
    ref_r = FusedDecodeRunner(model, mk(), use_graph=False)
    t, ref_ids = first, []
    for _ in range(12):
        lg = ref_r.step(t)
        t = torch.argmax(lg, -1)
        ref_ids.append(int(t.item()))
    for use_graph in (False, True):
        r = FusedDecodeRunner(model, mk(), use_graph=use_graph)
        r.start_greedy(first)
        ids = []
        for _ in range(12):
            lg = r.step_greedy()
            ids.append(int(torch.argmax(lg, -1).item()))
        assert ids == ref_ids, (use_graph, ids, ref_ids)
        assert r.caches[0].offset == 12


def test_fused_block_fp16_activations(oracle):
    """The fp16 instantiations of the fused launches (norm, residual, SwiGLU) and of attention / lm_head."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    from gbx_lm_amd.models.rope_utils import RoPE, apply_rope
    k, n, bits, g = 1024, 320, 4, 64
    def mod(seed):
        d = oracle.synth_layer(n, k, bits, g, seed=seed, act_dtype="f16")
        m = QuantizedLinear(k, n, False, g, bits)
        m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
        m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
        return m, d
    (gate, dg), (up, du) = mod(1), mod(2)
    x = torch.randn(k).to(torch.float16).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(torch.float16).to(DEV)
    act = torch.zeros(n, dtype=torch.float16, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(x, (gate, up), (act,), norm_weight=nw, eps=1e-5, swiglu=True), x.device)
    xn = _rms(x, nw, 1e-5).cpu()
    g64 = oracle.qlinear_ref(xn[None], dg["qweight"], dg["scales"], dg["zeros"], bits, g, acc=np.float64)
    u64 = oracle.qlinear_ref(xn[None], du["qweight"], du["scales"], du["zeros"], bits, g, acc=np.float64)
    oracle.assert_close(act[None], (g64 / (1 + np.exp(-g64))) * u64, "f16", "swiglu f16")
    h = torch.randn(n).to(torch.float16).to(DEV)
    h0 = h.clone()
    xa = torch.randn(k).to(torch.float16).to(DEV)
    ops.qlinear_fused(ops.make_fused_desc(xa, (gate,), (h,), residual=h), xa.device)
    y64 = oracle.qlinear_ref(xa.cpu()[None], dg["qweight"], dg["scales"], dg["zeros"], bits, g, acc=np.float64)
    oracle.assert_close(h[None], h0.cpu().double().numpy()[None] + y64, "f16", "residual f16")
    # attention + lm_head in fp16
    hd, nh, nkv, cap, pos = 128, 4, 2, 64, 9
    q, kk, v = (torch.randn(c * hd).to(torch.float16).to(DEV) for c in (nh, nkv, nkv))
    kc = torch.randn(1, nkv, cap, hd).to(torch.float16).to(DEV)
    vc = torch.randn(1, nkv, cap, hd).to(torch.float16).to(DEV)
    kc0, vc0 = kc.clone(), vc.clone()
    rope = RoPE(hd).to(DEV)
    out = torch.zeros(nh * hd, dtype=torch.float16, device=DEV)
    ops.attn_decode(q, kk, v, kc, vc, rope.inv_freq, torch.tensor([pos], device=DEV), out, nh, nkv, hd, cap, 1.0, hd ** -0.5, False)
    qr = apply_rope(q.view(1, nh, 1, hd), rope.inv_freq, pos, 1.0, False)
    kc0[:, :, pos:pos + 1] = apply_rope(kk.view(1, nkv, 1, hd), rope.inv_freq, pos, 1.0, False)
    vc0[:, :, pos:pos + 1] = v.view(1, nkv, 1, hd)
    K = kc0[:, :, :pos + 1].repeat_interleave(nh // nkv, 1).float()
    V = vc0[:, :, :pos + 1].repeat_interleave(nh // nkv, 1).float()
    want = (torch.softmax((qr.float() @ K.transpose(-1, -2)) * hd ** -0.5, -1) @ V).reshape(-1)
    assert torch.allclose(out.float(), want, atol=5e-3, rtol=5e-3)
    w = (torch.randn(777, k) * 0.02).to(torch.float16).to(DEV)
    y = ops.dense_gemv(x, w, norm_weight=nw, eps=1e-5)
    assert torch.allclose(y.double(), w.double() @ _rms(x, nw, 1e-5).double(), atol=2e-3, rtol=5e-3)


@pytest.mark.parametrize("bits,g", [(2, 128), (2, 32), (3, 64), (4, 128)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_packed_scales_zeros_companion(oracle, bits, g, dt):
    """gbx_pack_scales_zeros copies the 16-bit patterns verbatim, and a launch that reads the packed companion is
    bit-identical to the one that reads scales/zeros (same arithmetic, fewer loads): q/k/v, residual (KS=1 and the
    K-split long-row kernel) and SwiGLU."""
    from gbx_lm_amd import ops
    act = "bf16" if dt == torch.bfloat16 else "f16"
    def mod(n, k, seed):
        from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
        d = oracle.synth_layer(n, k, bits, g, seed=seed, act_dtype=act)
        m = QuantizedLinear(k, n, False, g, bits)
        m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
        m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
        return m
    k = 4096
    q, kk, v = mod(512, k, 1), mod(128, k, 2), mod(130, k, 3)
    pq = ops.pack_scales_zeros(q)
    s16, z16 = q.scales.view(torch.int16).int() & 0xFFFF, q.zeros.view(torch.int16).int() & 0xFFFF
    want = (s16 | (z16 << 16)).view(256, 2, k // g).transpose(1, 2).contiguous()
    assert torch.equal(pq, want)
    gate, up = mod(1030, k, 4), mod(1030, k, 5)
    pgu = ops.pack_scales_zeros(gate, up)
    sg, zg = gate.scales.view(torch.int16).int() & 0xFFFF, gate.zeros.view(torch.int16).int() & 0xFFFF
    su, zu = up.scales.view(torch.int16).int() & 0xFFFF, up.zeros.view(torch.int16).int() & 0xFFFF
    assert torch.equal(pgu, torch.stack([sg | (zg << 16), su | (zu << 16)], dim=-1))

    x = (torch.randn(k) * 2).to(dt).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(dt).to(DEV)
    def run_qkv(packed):
        ys = [torch.zeros(m.output_dims, dtype=dt, device=DEV) for m in (q, kk, v)]
        ops.qlinear_fused(ops.make_fused_desc(x, (q, kk, v), ys, norm_weight=nw, eps=1e-5, packed=packed), x.device)
        return ys
    for a, b in zip(run_qkv(None), run_qkv(torch.cat([ops.pack_scales_zeros(m) for m in (q, kk, v)]))):
        assert torch.equal(a, b)
    def run_gu(packed):
        o = torch.zeros(1030, dtype=dt, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (gate, up), (o,), norm_weight=nw, eps=1e-5, swiglu=True, packed=packed), x.device)
        return o
    assert torch.equal(run_gu(None), run_gu(pgu))
    for n, kd in ((4096, 14336), (130, 4096), (30000, 1024)):
        dn = mod(n, kd, 6)
        xd = torch.randn(kd).to(dt).to(DEV)
        h0 = torch.randn(n).to(dt).to(DEV)
        def run_dn(packed):
            h = h0.clone()
            ops.qlinear_fused(ops.make_fused_desc(xd, (dn,), (h,), residual=h, packed=packed), xd.device)
            return h
        assert torch.equal(run_dn(None), run_dn(ops.pack_scales_zeros(dn)))
    with pytest.raises(ValueError):
        odd = mod(129, k, 7)
        y = torch.zeros(129, dtype=dt, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (odd,), (y,), packed=torch.zeros((129 * (k // g),), dtype=torch.int32, device=DEV)), x.device)


@pytest.mark.experimental
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_tiled_gemv_fused_variants(oracle, dt):
    """Matrix-core GEMV with the decode-block fusions against the oracle and against the VALU fused launch:
    RMSNorm prologue + SwiGLU pair tiles (gate/up), residual epilogue on a long-row matrix (down_proj shape)."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    act = "bf16" if dt == torch.bfloat16 else "f16"
    def mod(n, k, seed):
        d = oracle.synth_layer(n, k, 2, 128, seed=seed, act_dtype=act)
        m = QuantizedLinear(k, n, False, 128, 2)
        m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
        m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
        return m, d
    k, n = 4096, 1032                       # n % 16 == 8: the last pair tile is full, the last plain tile is ragged
    (gate, dg), (up, du) = mod(n, k, 1), mod(n, k, 2)
    x = (torch.randn(k) * 2).to(dt).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(dt).to(DEV)
    wt, szt = ops.prepack_tiles(gate, up)
    act_t = ops.gemv_tiled(x, wt, szt, n, k, norm_weight=nw, eps=1e-5, swiglu=True)
    xn = _rms(x, nw, 1e-5).cpu()
    g64 = oracle.qlinear_ref(xn[None], dg["qweight"], dg["scales"], dg["zeros"], 2, 128, acc=np.float64)
    u64 = oracle.qlinear_ref(xn[None], du["qweight"], du["scales"], du["zeros"], 2, 128, acc=np.float64)
    oracle.assert_close(act_t[None], (g64 / (1 + np.exp(-g64))) * u64, act, "tiled swiglu")
    act_v = torch.zeros(n, dtype=dt, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(x, (gate, up), (act_v,), norm_weight=nw, eps=1e-5, swiglu=True), x.device)
    assert (act_t.float() - act_v.float()).abs().max().item() <= 0.02 * act_v.float().abs().max().item() + 1e-3
    # residual, K = 14336
    kd, nd = 14336, 520
    dn, dd = mod(nd, kd, 3)
    xd = torch.randn(kd).to(dt).to(DEV)
    h0 = torch.randn(nd).to(dt).to(DEV)
    wt2, szt2 = ops.prepack_tiles(dn)
    h = h0.clone()
    ops.gemv_tiled(xd, wt2, szt2, nd, kd, out=h, residual=h)
    y64 = oracle.qlinear_ref(xd.cpu()[None], dd["qweight"], dd["scales"], dd["zeros"], 2, 128, acc=np.float64)
    oracle.assert_close(h[None], h0.cpu().double().numpy()[None] + y64, act, "tiled residual")


@pytest.mark.parametrize("traditional", [False, True])
def test_rope_append_batched(traditional):
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import RoPE, apply_rope
    B, nh, nkv, hd, cap, pos = 5, 8, 2, 128, 64, 37
    q = torch.randn(B, nh * hd).to(torch.bfloat16).to(DEV)
    k = torch.randn(B, nkv * hd).to(torch.bfloat16).to(DEV)
    v = torch.randn(B, nkv * hd).to(torch.bfloat16).to(DEV)
    kc = torch.zeros(B, nkv, cap, hd, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    rope = RoPE(hd, traditional=traditional).to(DEV)
    q0 = q.clone()
    ops.rope_append(q, k, v, kc, vc, rope.inv_freq, torch.tensor([pos], device=DEV), nh, nkv, hd, 1.0, traditional)
    inv = rope.inv_freq.cpu()                       # reference: the torch fp32 formulation on the CPU
    qr = apply_rope(q0.cpu().view(B, 1, nh, hd).transpose(1, 2), inv, pos, 1.0, traditional).transpose(1, 2).reshape(B, -1)
    kr = apply_rope(k.cpu().view(B, 1, nkv, hd).transpose(1, 2), inv, pos, 1.0, traditional)
    assert torch.allclose(q.float().cpu(), qr.float(), atol=2e-2, rtol=2e-2)
    assert torch.allclose(kc[:, :, pos].float().cpu(), kr[:, :, 0].float(), atol=2e-2, rtol=2e-2)
    assert torch.equal(vc[:, :, pos], v.view(B, nkv, hd))
    assert torch.count_nonzero(kc[:, :, :pos]) == 0 and torch.count_nonzero(kc[:, :, pos + 1:]) == 0


@pytest.mark.parametrize("traditional", [False, True])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_rope_kernel_matches_torch_fp32(traditional, dt):
    """gbx_rope (one launch; what apply_rope runs on the GPU for every family) against the torch fp32 formulation on
    the CPU: strided (transposed projection) input, partial rotary dims, immediate and device-resident offsets."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import apply_rope
    g = torch.Generator().manual_seed(3)
    for (B, H, L, D, dims, off, scale) in ((2, 8, 17, 128, 128, 0, 1.0), (2, 6, 300, 64, 64, 2048, 1.0), (1, 4, 1, 64, 32, 4099, 0.25), (3, 2, 5, 96, 96, 120000, 1.0)):
        x = torch.randn(B, L, H, D, generator=g).to(dt)
        inv = 1.0 / (10000.0 ** (torch.arange(0, dims, 2, dtype=torch.float32) / dims))
        want = apply_rope(x.transpose(1, 2), inv, off, scale, traditional)
        xd = x.to(DEV).transpose(1, 2)                                   # strided view, as the model graphs pass it
        got = ops.rope(xd, inv.to(DEV), off, scale, traditional)
        got_dev = apply_rope(xd, inv.to(DEV), torch.tensor([off], device=DEV), scale, traditional)
        assert got.is_contiguous() and got.shape == want.shape
        # one bf16/fp16 rounding of an fp32 result on both sides; angles near 1e5 rad carry ~1e-2 of fp32 phase error
        tol = 4e-2 if off > 10000 else 1e-2
        assert torch.allclose(got.float().cpu(), want.float(), atol=tol, rtol=2e-2), (B, H, L, D, dims, off)
        assert torch.equal(got, got_dev)
        if dims < D:
            assert torch.equal(got[..., dims:].cpu(), x.transpose(1, 2)[..., dims:])
        if L >= 16:             # prompt-sized calls run rope_tokens_kernel (angles once per position) -- against the per-pair kernel:
            for t in (0, 7, L - 1):     # the same fp32 formula; the compiled sequences differ by an fp32 ulp on rare elements
                a, b = got[:, :, t:t + 1].float(), ops.rope(xd[:, :, t:t + 1], inv.to(DEV), off + t, scale, traditional).float()
                ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
                assert ((a - b).abs() <= ulp * b.abs().clamp_min(2.0 ** -14)).all() and (a != b).float().mean().item() < 2e-3


def test_rmsnorm_short_rows():
    """Per-head q / k norms (row length = head_dim): the wave-per-row variant of gbx_rmsnorm."""
    from gbx_lm_amd import ops
    for rows, k, dt in ((37, 128, torch.bfloat16), (4, 64, torch.float16), (1025, 512, torch.bfloat16)):
        x = torch.randn(rows, k).to(dt)
        w = (1 + 0.1 * torch.randn(k)).to(dt)
        xf = x.float()
        want = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float())
        got = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6).float().cpu()
        assert torch.allclose(got, want, atol=2e-2, rtol=1e-2), (rows, k)


def test_eminf_optimize_on_device(golden_dir, oracle):
    """gbx_eminf_optimize (one launch, decisions on the GPU) against the oracle's numpy restatement of
    gbx_lm/infer_opt.py:30-77 on the committed fixture, plus random vectors at vocabulary size."""
    import os
    from gbx_lm_amd.infer_opt import eminf_optimize
    z = np.load(os.path.join(golden_dir, "eminf.npz"))
    best, h = eminf_optimize(torch.from_numpy(z["logits"]).to(DEV))
    assert best.dtype == torch.float32 and best.shape == (512,)
    assert np.abs(best.cpu().numpy() - z["best"]).max() < 2e-3 and abs(float(h) - float(z["entropy"])) < 2e-3
    b2, h2 = eminf_optimize(torch.from_numpy(z["sharp"]).to(DEV))          # already below the target: input returned
    assert np.abs(b2.cpu().numpy() - z["sharp_best"]).max() < 1e-5 and abs(float(h2) - float(z["sharp_entropy"])) < 1e-4
    rng = np.random.default_rng(5)
    for v, scale, dt in ((128256, 3.0, torch.bfloat16), (32000, 1.5, torch.float32), (1000, 6.0, torch.float16)):
        lg = torch.from_numpy((rng.standard_normal(v) * scale).astype(np.float32)).to(dt)
        want, hw = oracle.eminf_optimize(lg.float().numpy())
        got, hg = eminf_optimize(lg.to(DEV))
        assert abs(float(hg) - float(hw)) < 5e-3 * max(1.0, float(hw)), (v, float(hg), float(hw))
        assert np.abs(got.cpu().numpy() - want).max() < 5e-3, (v, np.abs(got.cpu().numpy() - want).max())
    fixed, _ = eminf_optimize(torch.from_numpy(z["logits"]).to(DEV), num_steps=2)
    w2, _ = oracle.eminf_optimize(z["logits"], num_steps=2)
    assert np.abs(fixed.cpu().numpy() - w2).max() < 2e-3


def test_eminf_generate_step_on_device(tmp_path):
    """infer_opt.py:168-245 on the GPU: the generator form (prefill through the MFMA GEMMs, single-token steps through the
    graph-replayed decode block, logits optimised by gbx_eminf_optimize) yields the tokens eminf_generate returns under the
    same numpy stream, each with the logprob of the optimised distribution; a PromptCache hit feeds only the new tokens."""
    from gbx_lm_amd import synthetic as S, utils as U
    from gbx_lm_amd.infer_opt import eminf_generate, eminf_generate_step
    from gbx_lm_amd.prompt_cache import PromptCache
    S.write_checkpoint(str(tmp_path), S.TINY, S.make_strategy("S22", 2), seed=3)
    model, tok = U.load(str(tmp_path))
    model = model.to(DEV)
    for blk in model.model.layers:
        blk.self_attn.rope.to(DEV)
    prompt = [5, 17, 42, 9, 100, 3]
    ids = eminf_generate(model, tok, prompt, max_tokens=12, rng=np.random.default_rng(4))
    steps = list(eminf_generate_step(model, tok, prompt, max_tokens=12, rng=np.random.default_rng(4)))
    body = [t for t in ids if t not in tok.eos_token_ids]
    assert [t for t, _, _ in steps] == body[:len(steps)] and len(steps) >= 1
    for t, lp, hs in steps:
        assert hs is None and lp.is_cuda and float(lp) <= 0.0 and 0 <= t < 512
    pc = PromptCache()
    conv1, conv2 = prompt, prompt + [33, 34, 35, 36]
    eminf_generate(model, tok, conv1 + [4], conv1, max_tokens=2, prompt_cache=pc, rng=np.random.default_rng(5))
    for c in pc.cache:
        c.trim(c.offset - len(conv1))
    # turn 2 extends the conversation: the hit feeds only the new tokens and reproduces the from-scratch logits
    from gbx_lm_amd.infer_opt import get_input_ids
    from gbx_lm_amd.models.cache import make_prompt_cache
    todo, hit, cache = get_input_ids(pc, model, conv2 + [4], conv2, pc.model_key)
    assert hit and todo == conv2[len(conv1):] + [4] and pc.tokens_no_gen == conv2
    with torch.no_grad():
        got = model(torch.tensor([todo], device=DEV), cache=cache)[:, -1, :].float()
        want = model(torch.tensor([conv2 + [4]], device=DEV), cache=make_prompt_cache(model))[:, -1, :].float()
    assert torch.allclose(got, want, atol=5e-2, rtol=5e-2)
    steps2 = list(eminf_generate_step(model, tok, conv2 + [4, 7], conv2 + [4], max_tokens=4, prompt_cache=pc, rng=np.random.default_rng(6)))
    assert all(0 <= t < 512 for t, _, _ in steps2)


@pytest.mark.parametrize("bits,g", [(2, 128), (3, 128), (4, 128), (2, 64), (2, 32)])
@pytest.mark.parametrize("m", [2, 16, 32, 48])
def test_batched_swiglu_gemm(oracle, bits, g, m):
    """gbx_qlinear_swiglu_forward (gate | up + SwiGLU in the epilogue of the no-dequant batch GEMM) against the three unfused
    ops (same four roundings: equal up to one activation-dtype ulp) and against the oracle."""
    from gbx_lm_amd import ops
    gen = torch.Generator().manual_seed(1000 * bits + 10 * m + g)
    for n, k in ((14336, 4096), (1024, 2048), (8192, 3072)):
        (gate, dg), (up, du) = _mk(oracle, n, k, bits, g, seed=bits + m), _mk(oracle, n, k, bits, g, seed=bits + m + 77)
        x = torch.randn(m, k, generator=gen).to(torch.bfloat16).to(DEV)
        act = ops.quantized_swiglu(x, gate, up)
        assert act is not None and act.shape == (m, n)
        gg, uu = ops.quantized_matmul_multi(x, (gate, up))
        want = ops.swiglu(gg.contiguous(), uu.contiguous())
        # (not bit-identical: a workgroup starts its k walk at step blockIdx.x % steps, and the fused launch tiles the rows
        # differently, so the fp32 sums differ in their last bits and ~0.3 % of the outputs round the other way)
        d = (act.float() - want.float()).abs()
        rms = want.float().pow(2).mean().sqrt().item()
        # (rows beyond 32 of narrow matrices: the unfused path runs the dequantising kernel -- different arithmetic throughout)
        assert (d <= (4e-2 if m > 32 else 2e-2) * want.float().abs() + (4e-2 if m > 32 else 2e-3) * rms).all() and ((d > 0).float().mean().item() < 0.02 or m > 32), \
            f"N{n} K{k}: max diff {d.max().item()}, differing {(d > 0).float().mean().item():.4f}"
        rows = [0, m - 1]
        g64 = oracle.qlinear_ref(x[rows].cpu(), dg["qweight"], dg["scales"], dg["zeros"], bits, g, acc=np.float64)
        u64 = oracle.qlinear_ref(x[rows].cpu(), du["qweight"], du["scales"], du["zeros"], bits, g, acc=np.float64)
        oracle.assert_close(act[rows], (g64 / (1 + np.exp(-g64))) * u64, "bf16", f"batched swiglu b{bits} g{g} M{m}")


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("m", [2, 9, 32])
def test_partials_gemm_and_slice_folding_norm(oracle, bits, m):
    """gbx_qlinear_forward_partials + gbx_add_rmsnorm_slices (the summing launch of the k-split down projection folded into
    the next residual add + RMSNorm) against gbx_qlinear_forward_ws + gbx_add_rmsnorm: bit-identical h and y."""
    from gbx_lm_amd import ops
    n, k = 4096, 14336
    mod, d = _mk(oracle, n, k, bits, 128, seed=bits * 3 + m)
    x = torch.randn(m, k).to(torch.bfloat16).to(DEV)
    w = (1 + 0.1 * torch.randn(n)).to(torch.bfloat16).to(DEV)
    h0 = torch.randn(m, n).to(torch.bfloat16).to(DEV)
    parts = ops.quantized_matmul_partials(x, mod)
    assert parts is not None and parts[1] > 1
    h1 = h0.clone()
    y1 = ops.add_rmsnorm_slices(h1, parts[0], parts[1], w, 1e-5)
    h2 = h0.clone()
    delta = mod(x)
    y2 = ops.add_rmsnorm(h2, delta.contiguous(), w, 1e-5)
    assert torch.equal(h1, h2) and torch.equal(y1, y2)
    y64 = oracle.qlinear_ref(x[:1].cpu(), d["qweight"], d["scales"], d["zeros"], bits, 128, acc=np.float64)
    oracle.assert_close((h1[:1].float() - h0[:1].float()), y64, "bf16", "partials row 0")
    # o_proj-sized rows split too when the consumer folds the sum in; against the unsplit launch: equal up to the output rounding
    om, od = _mk(oracle, 4096, 4096, bits, 128, seed=bits + 40)
    xo = torch.randn(m, 4096).to(torch.bfloat16).to(DEV)
    po = ops.quantized_matmul_partials(xo, om)
    assert po is not None and po[1] > 1
    hz = torch.zeros(m, 4096, dtype=torch.bfloat16, device=DEV)
    ops.add_rmsnorm_slices(hz, po[0], po[1], w, 1e-5)
    y64 = oracle.qlinear_ref(xo.cpu(), od["qweight"], od["scales"], od["zeros"], bits, 128, acc=np.float64)
    oracle.assert_close(hz, y64, "bf16", "o_proj partials")
    small, _ = _mk(oracle, 512, 1024, bits, 128, seed=1)
    assert ops.quantized_matmul_partials(torch.randn(4, 1024).to(torch.bfloat16).to(DEV), small) is None     # short rows do not split


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [(32, 128256, 4096), (2, 1000, 512), (300, 4096 + 40, 3072), (1, 256, 64),
                                   (7, 32768 + 100, 1024), (17, 151936, 2048), (32, 32000 + 1024, 128)])   # M <= 32, N >= 32768: the row-streaming kernel
def test_dense_forward_matches_fp64(dtype, m, n, k):
    """gbx_dense_forward (the B > 1 output head / the GEMM stage of the long-prompt path) against an fp64 matmul of the same
    16-bit operands, incl. ragged M / N and a bias."""
    from gbx_lm_amd import ops
    x = torch.randn(m, k, device=DEV).to(dtype)
    w = (torch.randn(n, k, device=DEV) * 0.02).to(dtype)
    b = torch.randn(n, device=DEV).to(dtype) if n % 3 == 1 else None
    y = ops.dense_linear(x, w, b)
    ref = x.double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    rms = ref.pow(2).mean().sqrt().item()
    assert y.shape == (m, n) and ((y.double() - ref).abs() <= tol * rms + tol * ref.abs()).all()


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,slices", [(32, 4096, 8), (5, 4096, 3), (32, 3072, 7), (17, 2048, 1), (40, 3072, 2), (64, 2048, 4), (1, 512, 2)])
def test_chip_wide_slice_norm_vs_the_row_per_workgroup_launch(dt, m, k, slices):
    """gbx_add_rmsnorm_slices_xcd (a row cut into 512-value segments over the whole chip, the statistic exchanged inside one XCD's L2)
    against gbx_add_rmsnorm_slices: h bit-identical (same slice order and roundings), y equal up to one place of the output type (the
    statistic is summed per segment), and y against an fp64 restatement.  One state buffer serves 40 launches back to back with
    changing inputs (generation tags), and the status word stays 0."""
    from gbx_lm_amd import ops
    gen = torch.Generator().manual_seed(m * 7 + k + slices)
    state = ops.norm_state(DEV)
    w = (1 + 0.1 * torch.randn(k, generator=gen)).to(dt).to(DEV)
    for it in range(40):
        ws = (torch.randn(slices, m, k, generator=gen) * (0.5 + 0.1 * it)).to(DEV)
        h0 = torch.randn(m, k, generator=gen).to(dt).to(DEV)
        h1, h2 = h0.clone(), h0.clone()
        y1 = ops.add_rmsnorm_slices(h1, ws, slices, w, 1e-5, state)
        y2 = ops.add_rmsnorm_slices(h2, ws, slices, w, 1e-5)
        assert torch.equal(h1, h2), f"launch {it}: the residual stream differs"
        ulp = 2.0 ** (-7 if dt == torch.bfloat16 else -10)
        assert ((y1.double() - y2.double()).abs() <= ulp * y2.double().abs() + 1e-30).all(), f"launch {it}: y differs by more than one place"
    assert ops.norm_status(state) == 0
    acc = ws[0].clone()
    for s in range(1, slices):
        acc += ws[s]
    hh = (h0.float() + acc.to(dt).float()).to(dt)
    assert torch.equal(h1, hh)
    hf = hh.double()
    want = hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5) * w.double()
    err = (y1.double() - want).abs()
    assert (err <= 2 * ulp * want.abs() + 1e-6).all()


def test_chip_wide_slice_norm_refuses_what_it_does_not_cover():
    from gbx_lm_amd import _lib, ops
    state = ops.norm_state(DEV)
    for m, k in ((33, 4096), (4, 4608), (4, 768), (41, 3072)):
        h = torch.zeros(m, k, dtype=torch.bfloat16, device=DEV)
        ws = torch.zeros(1, m, k, device=DEV)
        rc = _lib.lib().gbx_add_rmsnorm_slices_xcd(h.data_ptr(), ws.data_ptr(), 1, h.data_ptr(), h.data_ptr(), m, k, 1e-5, _lib.GBX_BF16, state.data_ptr(), None)
        assert rc == _lib.GBX_ERR_UNSUPPORTED, (m, k, rc)
        # the host wrapper keeps such shapes on the row-per-workgroup launch
        y = ops.add_rmsnorm_slices(h, ws, 1, torch.ones(k, dtype=torch.bfloat16, device=DEV), 1e-5, state)
        assert y.shape == h.shape
