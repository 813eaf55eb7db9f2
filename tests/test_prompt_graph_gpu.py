"""decode.PromptRunner: the graph-replayed short-prompt pass against the eager pass (gbx_lm/utils.py:312-323) -- logits of the
last position, cache contents and the tokens generate_step yields afterwards."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model():
    from gbx_lm_amd import synthetic as S
    cfg = dict(S.TINY, hidden_size=1024, intermediate_size=2048, num_attention_heads=8, num_key_value_heads=2, num_hidden_layers=3)
    return S.build_model(cfg, S.make_strategy("S22", 3, 128), device=DEV, seed=11), cfg


@pytest.mark.parametrize("L", [2, 31, 32, 33, 100, 128])
def test_prompt_graph_matches_eager(L):
    from gbx_lm_amd.decode import PromptRunner, prompt_step
    from gbx_lm_amd.models.cache import make_prompt_cache
    model, cfg = _model()
    torch.manual_seed(L)
    y = torch.randint(1, cfg["vocab_size"], (L,), device=DEV)
    bucket = next(b for b in PromptRunner.BUCKETS if b >= L)
    # eager reference on the SAME padded rows (tile dispatch depends on the row count), last position picked afterwards
    ids = torch.zeros((1, bucket), dtype=torch.long, device=DEV)
    ids[0, :L] = y
    ref_cache = make_prompt_cache(model)
    with torch.no_grad():
        hidden = model.model(ids, None, ref_cache)
        want = torch.nn.functional.linear(hidden[:, L - 1:L, :], model.head_weight())[:, -1, :]
    cache = make_prompt_cache(model)
    got = prompt_step(model, y, cache)
    assert got is not None
    assert torch.equal(got, want)
    for c, r in zip(cache, ref_cache):
        assert c.offset == L
        assert torch.equal(c.keys[:, :, :L], r.keys[:, :, :L]) and torch.equal(c.values[:, :, :L], r.values[:, :, :L])
    # ... and against the un-padded eager pass within the accumulation-order tolerance of a different tile choice
    plain = make_prompt_cache(model)
    with torch.no_grad():
        lg = model.last_logits(y[None], plain)
    assert torch.allclose(got.float(), lg.float(), atol=0.05 * float(lg.float().std()) + 0.02, rtol=0.05)
    # a second call reuses the bucket's graph; a non-empty cache falls back
    cache2 = make_prompt_cache(model)
    assert torch.equal(prompt_step(model, y, cache2), want)
    assert prompt_step(model, y, cache2) is None
    assert prompt_step(model, torch.randint(1, 100, (200,), device=DEV), make_prompt_cache(model)) is None      # GPU-bound sizes: eager


def test_generate_step_uses_the_graph_and_agrees_with_eager():
    from gbx_lm_amd.utils import generate_step
    model, cfg = _model()
    y = torch.randint(1, cfg["vocab_size"], (100,), device=DEV)

    def toks(flag):
        os.environ["GBX_PROMPT_GRAPH"] = flag
        try:
            out = []
            for t, lp, _ in generate_step(y, model, max_tokens=6):
                out.append((t, lp.float().cpu()))
            return out
        finally:
            os.environ.pop("GBX_PROMPT_GRAPH", None)
    a, b = toks("1"), toks("0")
    assert "_prompt_runners" in model.__dict__ and 128 in model._prompt_runners
    assert a[0][0] == b[0][0]
    assert torch.allclose(a[0][1], b[0][1], atol=0.1, rtol=0.05)
