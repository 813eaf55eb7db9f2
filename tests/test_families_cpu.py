"""CPU: the other dense families of the reference registry that reuse the path with small graph deltas:
Qwen2 (biased q/k/v), Qwen3 (per-head q/k RMSNorm), Mixtral (unquantised router, stacked experts named w1/w2/w3)
and Phi-3 (fused qkv / gate_up projections, su-scaled RoPE) -- structure, loader contract, generate loop through the oracle."""
import pytest
import torch

from gbx_lm_amd import synthetic as S, utils as U
from gbx_lm_amd.models import qqwen2, qqwen3


def test_registry_and_structure():
    assert U.MODEL_MAPPING["qwen2"] is qqwen2 and U.MODEL_MAPPING["qwen3"] is qqwen3
    m2 = S.build_model(S.TINY_QWEN2, S.make_strategy("S22", 2, 64), device="cpu", seed=1)
    at = m2.model.layers[1].self_attn
    assert all(getattr(at, p).bias is not None for p in ("q_proj", "k_proj", "v_proj")) and at.o_proj.bias is None   # qqwen2.py:44-47
    assert m2.model.layers[0].mlp.gate_proj.bias is None and not hasattr(m2, "lm_head")                              # tied head
    assert at.rope.inv_freq[1].item() == pytest.approx(1e6 ** (-2 / 64))                                            # theta = 1e6
    m3 = S.build_model(S.TINY_QWEN3, S.make_strategy("S22", 2, 64), device="cpu", seed=1)
    names = dict(m3.named_parameters()).keys()
    assert "model.layers.0.self_attn.q_norm.weight" in names and "model.layers.1.self_attn.k_norm.weight" in names
    assert m3.model.layers[0].self_attn.head_dim == 64 and hasattr(m3, "lm_head")
    from gbx_lm_amd.models import qmixtral
    from gbx_lm_amd.models.switch_layers import QuantizedSwitchLinear
    mx_ = S.build_model(S.TINY_MIXTRAL, S.make_strategy("S22", 2, 64), device="cpu", seed=1)
    moe = mx_.model.layers[0].block_sparse_moe
    assert U.MODEL_MAPPING["mixtral"] is qmixtral and isinstance(moe.gate, torch.nn.Linear)        # qmixtral.py:108
    assert isinstance(moe.switch_mlp.down_proj, QuantizedSwitchLinear) and moe.switch_mlp.down_proj.bias is None
    from gbx_lm_amd.models import qphi3
    from gbx_lm_amd.models.rope_utils import SuScaledRotaryEmbedding
    strat = S.make_strategy("S22", 2, 64)
    for blk in ("model.layers.0", "model.layers.1"):            # a Phi-3 strategy file has only the fused entries
        for k in ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj"):
            del strat[blk][k]
    strat["model.layers.1"]["qkv_proj"] = {"bits": [2], "group_size": {"2": 32}}
    p3 = S.build_model(S.TINY_PHI3, strat, device="cpu", seed=1)
    at = p3.model.layers[1].self_attn
    assert U.MODEL_MAPPING["phi3"] is qphi3 and isinstance(at.rope, SuScaledRotaryEmbedding)
    assert at.qkv_proj.output_dims == 256 + 2 * 128 and (at.qkv_proj.bits, at.qkv_proj.group_size) == (2, 32)
    assert p3.model.layers[0].mlp.gate_up_proj.output_dims == 1024
    assert at.rope.scale == pytest.approx((1 + __import__("math").log(4096 / 512) / __import__("math").log(512)) ** 0.5)
    with pytest.raises(ValueError):
        qphi3.ModelArgs.from_dict(dict(S.TINY_PHI3, rope_scaling={"type": "su"}))          # long_factor missing (qphi3.py:34-37)
    with pytest.raises(ValueError):
        U._get_classes({"model_type": "gpt_oss"})                  # not built: fails like the reference does (utils.py:153-156)


@pytest.mark.parametrize("cfg", [S.TINY_QWEN2, S.TINY_QWEN3, S.TINY_MIXTRAL, S.TINY_PHI3], ids=["qwen2", "qwen3", "mixtral", "phi3"])
def test_checkpoint_roundtrip_and_generate(tmp_path, cpu_oracle_op, cfg):
    strat = S.make_strategy("S22", 2, 64)
    extra = {"lm_head.weight": torch.zeros((cfg["vocab_size"], cfg["hidden_size"]))} if cfg.get("tie_word_embeddings") else None
    src = S.write_checkpoint(str(tmp_path), cfg, strat, seed=6, extra_tensors=extra)
    model, tok = U.load(str(tmp_path))
    model = model.to("cpu")
    a, b = S.state_dict_for_disk(src), S.state_dict_for_disk(model)
    assert set(a) == set(b)
    for k in a:
        ta, tb = a[k], b[k]
        if ta.dtype == torch.uint32:
            ta, tb = ta.view(torch.int32), tb.view(torch.int32)
        assert torch.equal(ta.to(tb.dtype), tb), k
    prompt = torch.tensor([9, 8, 7, 6])
    want = [t for t, _, _ in U.generate_step(prompt, src, max_tokens=5)]
    got = [t for t, _, _ in U.generate_step(prompt, model, max_tokens=5)]
    assert got == want and len(got) == 5
    # the incremental (cached) pass equals the full-sequence pass: causal mask + cache offset + per-head norms / biases
    seq = torch.tensor([list(prompt) + want[:-1]])
    with torch.no_grad():
        full = src(seq)[0, len(prompt) - 1:, :].float()
    lps = torch.stack([lp for _, lp, _ in U.generate_step(prompt, src, max_tokens=5)]).float()
    full_lp = full - torch.logsumexp(full, -1, keepdim=True)
    assert (full_lp - lps).abs().max().item() < 0.08
