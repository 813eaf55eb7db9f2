"""GPU: gbx_attn_prefill (csrc/attn_prefill.hip), the L > 1 causal attention of the prompt pass, against softmax(q k^T * scale +
mask) v in fp32 with the mask of gbx_lm/models/base.py:27-62 (create_causal_mask: query i of the chunk sees keys 0 .. offset + i)
-- the call the reference makes at gbx_lm/models/base.py:101-122 from qllama.py:84-95 -- and, through the model graph, against
the torch SDPA route it displaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(q, k, v, offset, scale):
    b, hq, l, d = q.shape
    rep = hq // k.shape[1]
    kf, vf = k.float().repeat_interleave(rep, 1), v.float().repeat_interleave(rep, 1)
    s = q.float() @ kf.transpose(-1, -2) * scale
    i = torch.arange(l, device=q.device)[:, None] + offset
    j = torch.arange(k.shape[2], device=q.device)[None]
    s = s.masked_fill(j > i, float("-inf"))
    return (torch.softmax(s, -1) @ vf).permute(0, 2, 1, 3).reshape(b, l, hq * d)


def _case(b, hq, hkv, l, offset, d, dtype, seed=0, cache_cap=None, amp=1.0):
    from gbx_lm_amd import ops
    g = torch.Generator(device=DEV)
    g.manual_seed(seed + l + offset)
    s = offset + l
    q = (torch.randn((b, l, hq, d), device=DEV, generator=g) * amp).to(dtype).permute(0, 2, 1, 3)      # the _heads view of a projection
    cap = cache_cap or s
    kc = torch.randn((b, hkv, cap, d), device=DEV, generator=g).to(dtype)
    vc = torch.randn((b, hkv, cap, d), device=DEV, generator=g).to(dtype)
    k, v = kc[..., :s, :], vc[..., :s, :]                                                              # cache slices (KVCache.update_and_fetch)
    assert ops.attn_prefill_supported(q, k, v)
    out = ops.attn_prefill(q, k, v, offset, d ** -0.5)
    torch.cuda.synchronize()
    ref = _ref(q, k, v, offset, d ** -0.5)
    assert out.shape == (b, l, hq * d) and torch.isfinite(out.float()).all()
    err = (out.float() - ref).abs()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # P is rounded to the activation dtype before P V (relative eps per term, terms non-negative weights of |v| <~ 4) and the
    # output once more: a few eps of the row's magnitude scale
    bound = 4 * eps * ref.abs() + 6 * eps
    assert (err <= bound).all(), f"B{b} Hq{hq} Hkv{hkv} L{l} off{offset} D{d} {dtype}: max err {err.max().item():.4g}, {(err > bound).sum().item()} outside"


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_shapes(d, dtype):
    for l, off in [(2, 0), (17, 0), (64, 0), (128, 0), (129, 0), (200, 7), (1000, 0), (33, 300), (512, 512), (5, 1023)]:
        _case(2, 8, 2, l, off, d, dtype)
    _case(1, 4, 4, 300, 11, d, dtype)                      # multi-head (no grouping)
    _case(3, 6, 1, 77, 0, d, dtype)                        # multi-query
    _case(1, 8, 2, 260, 40, d, dtype, cache_cap=1024)      # cache with spare capacity: batch / head strides differ from S * D


def test_sharp_rows():
    """large logits: one key dominates, the running maximum moves between key tiles"""
    _case(1, 8, 2, 700, 0, 128, torch.bfloat16, amp=6.0)
    _case(1, 8, 2, 300, 64, 64, torch.float16, amp=4.0)


def test_cfg4_shape():
    """BASELINE.json configs[3]: 8 prompts x 4096 tokens, 32 query / 8 kv heads of 128"""
    _case(2, 32, 8, 4096, 0, 128, torch.bfloat16)


def test_rejects():
    from gbx_lm_amd import _lib, ops
    q = torch.randn((1, 4, 16, 96), device=DEV).to(torch.bfloat16)
    assert not ops.attn_prefill_supported(q, q, q)
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty((1, 16, 4 * 96), dtype=torch.bfloat16, device=DEV)
    assert L.gbx_attn_prefill(q.data_ptr(), q.data_ptr(), q.data_ptr(), out.data_ptr(), 1, 4, 4, 16, 16, 96, 4 * 16 * 96, 16 * 96, 96,
                              4 * 16 * 96, 16 * 96, 4 * 16 * 96, 16 * 96, 0, 0.1, _lib.GBX_BF16, st) == _lib.GBX_ERR_UNSUPPORTED
    assert L.gbx_attn_prefill(q.data_ptr(), q.data_ptr(), q.data_ptr(), out.data_ptr(), 1, 4, 4, 16, 8, 128, 0, 0, 128, 0, 0, 0, 0, 0, 0.1,
                              _lib.GBX_BF16, st) == _lib.GBX_ERR_INVALID          # S < offset + L


def test_model_prefill_uses_it_and_agrees_with_sdpa(monkeypatch):
    """chunked prompt pass of a model (second chunk at offset > 0) through the HIP attention vs the torch SDPA route"""
    from gbx_lm_amd import ops, synthetic as S
    from gbx_lm_amd.models.cache import make_prompt_cache
    cfg = dict(S.TINY)
    model = S.build_model(cfg, S.make_strategy("S40", cfg["num_hidden_layers"], 64), device=torch.device(DEV), seed=1)
    toks = torch.randint(0, cfg["vocab_size"], (2, 150), device=DEV)
    calls = []
    real = ops.attn_prefill

    def spy(*a, **kw):
        calls.append(a[3])
        return real(*a, **kw)
    monkeypatch.setattr(ops, "attn_prefill", spy)

    def run():
        cache = make_prompt_cache(model)
        model.prefill(toks[:, :100], cache)
        return model.last_logits(toks[:, 100:], cache).float()
    a = run()
    n_layers = cfg["num_hidden_layers"]
    assert calls == [0] * n_layers + [100] * n_layers
    monkeypatch.setattr(ops, "attn_prefill_supported", lambda *a: False)
    b = run()
    assert len(calls) == 2 * n_layers
    rms = b.pow(2).mean().sqrt().item()
    assert (a - b).abs().max().item() <= 0.05 * rms, f"max diff {(a - b).abs().max().item()} (rms {rms})"


def test_prompt_pass_residual_adds_inside_the_norms(monkeypatch):
    """LlamaModel's prompt-sized forward folds every residual add into the next RMSNorm launch (gbx_add_rmsnorm): same
    roundings as the op-by-op loop, so the hidden states are identical bit for bit"""
    from gbx_lm_amd import synthetic as S
    from gbx_lm_amd.models import qllama
    from gbx_lm_amd.models.cache import make_prompt_cache
    cfg = dict(S.TINY, num_hidden_layers=3, hidden_size=1024, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2)
    model = S.build_model(cfg, S.make_strategy("S22", 3, 64), device=torch.device(DEV), seed=3)
    assert model.model._fused_residuals(torch.empty((2, 70, 1024), dtype=torch.bfloat16, device=DEV))
    toks = torch.randint(0, cfg["vocab_size"], (2, 70), device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    a = model.model(toks, None, make_prompt_cache(model))
    monkeypatch.setattr(qllama.LlamaModel, "_fused_residuals", lambda self, h: False)
    b = model.model(toks, None, make_prompt_cache(model))
    assert torch.equal(a, b)


@pytest.mark.parametrize("traditional", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rope_keys_written_straight_into_the_cache(traditional, dtype):
    """rope(k) + cache.update_and_fetch(k, v) of a prompt chunk as one launch (gbx_rope_kv_write) against the two-step route: the
    same cache contents and returned views bit for bit, over two chunks with a cache growth in between"""
    from gbx_lm_amd.models.base import rope_keys_and_update_cache
    from gbx_lm_amd.models.cache import KVCache
    from gbx_lm_amd.models.rope_utils import RoPE
    B, H, D = 2, 4, 128
    rope = RoPE(D, traditional=traditional, base=10000.0).to(DEV)
    fused, plain = KVCache(step=64), KVCache(step=64)
    start = 0
    for L in (40, 24, 300):
        k = torch.randn((B, L, H, D), device=DEV).to(dtype).permute(0, 2, 1, 3)          # the _heads view of a projection output
        v = torch.randn((B, L, H, D), device=DEV).to(dtype).permute(0, 2, 1, 3)
        kf, vf = rope_keys_and_update_cache(rope, k, v, fused, start)
        kp, vp = plain.update_and_fetch(rope(k, offset=start), v)
        start += L
        assert fused.offset == plain.offset == start and kf.shape == kp.shape == (B, H, start, D)
        assert torch.equal(kf, kp) and torch.equal(vf, vp)
    # shapes the launch does not take (a short chunk) go through the two steps and stay consistent
    k = torch.randn((B, 3, H, D), device=DEV).to(dtype).permute(0, 2, 1, 3)
    v = torch.randn((B, 3, H, D), device=DEV).to(dtype).permute(0, 2, 1, 3)
    kf, vf = rope_keys_and_update_cache(rope, k, v, fused, start)
    kp, vp = plain.update_and_fetch(rope(k, offset=start), v)
    assert torch.equal(kf, kp) and torch.equal(vf, vp)
