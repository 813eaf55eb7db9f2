"""GPU: generate_step's default-sampler loop as ONE graph replay per token (FusedDecodeRunner.step_gen) against the step-by-step path.

Reference loop: gbx_lm/utils.py:217-338 -- ``_step`` = model call (:293-303), ``logprobs = logits - logsumexp(logits)`` (:305), default
sampler ``argmax(logprobs)`` (:285), one-token look-ahead with ``async_eval`` (:323-338).  The fast path runs the blocks, the head, the
logprobs and the argmax inside the captured graph, feeds the token back on the device and sends (token, hand-off status words) to the host in
one copy; a custom sampler (any callable) keeps the step-by-step path: decode_step -> logprobs -> sampler.  Both must yield the same tokens
and bit-identical logprobs, across a growth of the KV cache (a rebuilt runner re-seeds the device-side feedback) and for a cache handed in
by the caller."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(layers=3, bits=(2, 4, 3), seed=5):
    from gbx_lm_amd import synthetic as S
    cfg = dict(S.LLAMA3_8B, num_hidden_layers=layers, vocab_size=8192)
    strat = S.make_strategy("S22", layers, 128)
    for li, b in enumerate(bits):
        for p in strat["model.layers.%d" % li]:
            strat["model.layers.%d" % li][p] = {"bits": [b], "group_size": {str(b): 128}}
    return S.build_model(cfg, strat, device=DEV, seed=seed)


def _collect(gen):
    toks, lps = [], []
    for tok, lp, _ in gen:
        toks.append(tok)
        lps.append(lp.float().cpu().numpy().copy())
    return toks, lps


# (7, 24): a short prompt; (250, 20): the step-by-step loop's cache grows at 256; (370, 40): the one-launch block's context limit (384 cached
# positions: the replayed graph changes to the two-launch block mid-generation); (1010, 30): the runner's cache is full at 1024 and doubles
@pytest.mark.parametrize("prompt_len,ntok", [(7, 24), (250, 20), (370, 40), (1010, 30)])
def test_one_replay_per_token_equals_the_step_by_step_loop(prompt_len, ntok):
    from gbx_lm_amd import utils
    from gbx_lm_amd.decode import FusedDecodeRunner
    model = _model()
    prompt = (torch.arange(prompt_len) * 37 + 11) % 8000
    fast_t, fast_lp = _collect(utils.generate_step(prompt, model, max_tokens=ntok))
    r = model._decode_runner
    assert isinstance(r, FusedDecodeRunner) and any(k[0] == "gen" for k in r._graphs), "the one-replay path did not run"
    assert r._gen_next is not None and r._gen_next[0] == r.caches[0].offset
    if prompt_len == 370:                                     # both graphs of the one-replay mode ran: one-launch block, then two launches
        assert r.db[0] is not None and ("gen", False) in r._graphs and ("gen", True) in r._graphs and r.caches[0].offset > r.db_max_ctx
    if prompt_len == 1010:
        assert r.capacity == 2048
    model._decode_runner = None
    slow_t, slow_lp = _collect(utils.generate_step(prompt, model, max_tokens=ntok, sampler=lambda x: torch.argmax(x, dim=-1)))
    assert not any(k[0] == "gen" for k in model._decode_runner._graphs)
    assert fast_t == slow_t
    for a, b in zip(fast_lp, slow_lp):
        assert np.array_equal(a, b)
    # a yielded logprobs tensor stays what it was when the next step has run (the static buffer is copied per token)
    assert len({id(x) for x in fast_lp}) == len(fast_lp)
    assert all(abs(np.exp(a.astype(np.float64)).sum() - 1.0) < 0.05 for a in fast_lp)


def test_caller_interleaving_plain_steps_reseeds_the_feedback():
    """step_gen after a plain step() (or with a token tensor it did not hand out) takes token and position from the host again."""
    from gbx_lm_amd import utils
    from gbx_lm_amd.decode import decode_step, decode_step_logprobs
    from gbx_lm_amd.models.cache import make_prompt_cache
    model = _model(seed=9)
    prompt = (torch.arange(12) * 5 + 3) % 8000
    ref_t, ref_lp = _collect(utils.generate_step(prompt, model, max_tokens=6, sampler=lambda x: torch.argmax(x, dim=-1)))
    cache = make_prompt_cache(model)
    assert list(utils.generate_step(prompt, model, max_tokens=0, prompt_cache=cache)) == [] and cache[0].offset == prompt.numel()
    y = torch.tensor([ref_t[0]], device=DEV)
    got = [ref_t[0]]
    # gen, gen (fed back), plain step, gen (re-seeded from a host-built tensor), gen
    for mode in ("gen", "gen", "plain", "gen", "gen"):
        if mode == "gen":
            tok, lp, words = decode_step_logprobs(model, y, cache)
            assert int(words[0]) == int(tok) and not words[1:].any()
            y = tok if len(got) % 2 else torch.tensor([int(tok)], device=DEV)        # sometimes the handed-out view, sometimes a fresh tensor
            got.append(int(tok))
        else:
            logits = decode_step(model, y, cache)
            y = torch.argmax(logits, -1).reshape(-1)
            got.append(int(y))
    assert got == ref_t[:len(got)]
