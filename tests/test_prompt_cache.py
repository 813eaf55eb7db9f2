"""Host logic of the conversation prompt-prefix cache (gbx_lm/prompt_cache.py:84-147): the four decisions, and that a
cache hit yields the same logits as processing the whole conversation from scratch (CPU, oracle-backed tiny model)."""
import pytest
import torch


class _Tok:
    """Deterministic stand-in for a chat template: role tag + one id per character."""
    def apply_chat_template(self, messages, add_generation_prompt=False, enable_thinking=False):
        out = []
        for m in messages:
            out += [{"system": 1, "user": 2, "assistant": 3}[m["role"]]] + [10 + (ord(c) % 400) for c in m["content"]]
        return out + ([4] if add_generation_prompt else [])


@pytest.fixture
def model(cpu_oracle_op):
    from gbx_lm_amd import synthetic as S
    return S.build_model(S.TINY, S.make_strategy("S22", 2, 64), device="cpu", seed=5)


def _logits(model, cache, toks):
    with torch.no_grad():
        return model(torch.tensor([toks]), cache=cache)[:, -1, :]


def test_decisions_and_logit_equivalence(model):
    from gbx_lm_amd.prompt_cache import PromptCache
    from gbx_lm_amd.models.cache import make_prompt_cache
    tok = _Tok()
    pc = PromptCache()
    pc.cache_system_prompt(model, "be brief", tok)
    sys_t = list(pc.system_tokens)
    assert pc.system_cached and pc.cache[0].offset == len(sys_t)
    # turn 1: extends the cached system prompt -> only the new tokens are processed (case 4)
    conv1 = sys_t + tok.apply_chat_template([{"role": "user", "content": "hello"}])
    with_gen1 = conv1 + [4]
    todo, cache, hit = pc.get_prompt_cache(model, with_gen1, conv1, pc.model_key)
    assert hit and todo == with_gen1[len(sys_t):] and cache is pc.cache
    got = _logits(model, cache, todo)
    want = _logits(model, make_prompt_cache(model), with_gen1)
    assert torch.allclose(got.float(), want.float(), atol=2e-2, rtol=2e-2)
    # a different user turn after the same system prompt (case 3, kept): cache trimmed to the shared prefix
    conv2 = sys_t + tok.apply_chat_template([{"role": "user", "content": "hexagon"}])
    todo, cache, hit = pc.get_prompt_cache(model, conv2 + [4], conv2, pc.model_key)
    shared = len(sys_t) + 3                                   # role tag + "he"
    assert hit and todo == (conv2 + [4])[shared:] and cache[0].offset == shared
    got = _logits(model, cache, todo)
    want = _logits(model, make_prompt_cache(model), conv2 + [4])
    assert torch.allclose(got.float(), want.float(), atol=2e-2, rtol=2e-2)
    # a request that does not start with the cached system prompt (case 2) and another model (case 1): fresh caches
    other = tok.apply_chat_template([{"role": "user", "content": "x"}])
    todo, cache, hit = pc.get_prompt_cache(model, other + [4], other, pc.model_key)
    assert not hit and todo == other + [4] and cache[0].offset == 0 and not pc.system_cached
    todo, cache, hit = pc.get_prompt_cache(model, other + [4], other, "another-model")
    assert not hit and pc.model_key == "another-model"
    # divergence without a system prompt cached: start over (case 3, dropped)
    pc.tokens_no_gen = other + [7, 8]
    todo, cache, hit = pc.get_prompt_cache(model, other + [9, 4], other + [9], "another-model")
    assert not hit and todo == other + [9, 4]
    pc.update_after_step("ok", tok)
    assert pc.tokens_no_gen[-3:] == tok.apply_chat_template([{"role": "assistant", "content": "ok"}])
    with pytest.raises(NotImplementedError):
        PromptCache(quantize=True, qbit=4, q_group_size=64)
