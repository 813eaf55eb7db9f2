"""EMINF (entropy minimisation at inference) decode loop: second caller of the batch-1 QuantizedLinear path.

Host-side counterpart of gbx_lm/infer_opt.py: ``eminf_optimize`` (:30-77, the logits-only normalised-gradient
descent on entropy) and ``eminf_generate`` (:79-142, prefill, then per token: optimise logits, sample with numpy,
single-token model call).  The logits-side math is a few [vocab] vector ops in fp32 torch; every model call goes through
the same HIP kernels as ``generate`` (single-token steps use ``model.decode_step``).  ``get_input_ids`` (:10-28) hooks the
conversation prompt-prefix cache (``gbx_lm_amd.prompt_cache.PromptCache``) in front of the prefill; ``eminf_generate_step``
(:168-245) is the generator form the reference server consumes (fastapi_server.py:1313-1349).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from .models.cache import make_prompt_cache
from .tokenizer_utils import TokenizerWrapper


def _entropy(x: torch.Tensor) -> torch.Tensor:
    probs = torch.softmax(x, dim=-1)
    return -torch.sum(probs * torch.log(probs + 1e-10))


def _step_allocation(h_init: float, h_target: float, min_step: int = 3, max_step: int = 15) -> int:
    r = max(0.0, h_init - h_target) / h_target
    s = r / (1.0 + r)
    return int(min_step + max_step * s)


def eminf_optimize(logits: torch.Tensor, alpha: float = 0.65, num_steps: Optional[int] = None, threshold: float = 0.05):
    """infer_opt.py:30-77.  Returns (best_logits fp32, best_entropy 0-d tensor).  Device tensors run the whole optimisation
    as ONE HIP launch with every accept / reject decision on the GPU (gbx_eminf_optimize); the loop below is the host-side
    statement of the same algorithm (CPU tensors), with the reference's one scalar read-back per decision."""
    if logits.is_cuda:
        from . import ops
        best, ent = ops.eminf_optimize(logits, alpha=alpha, num_steps=num_steps, threshold=threshold)
        return best.reshape(logits.shape), ent[0]
    current_logits = logits.to(torch.float32)
    initial_alpha = alpha
    initial_entropy = _entropy(current_logits)
    best_entropy = initial_entropy
    best_logits = current_logits
    target_entropy = max(threshold, 0.1 * float(initial_entropy))
    if num_steps is None:
        num_steps = _step_allocation(float(initial_entropy), threshold)
    for _ in range(num_steps):
        current_entropy = _entropy(current_logits)
        if float(current_entropy) < target_entropy:
            break
        current_logp = torch.log_softmax(current_logits, dim=-1)
        current_p = torch.exp(current_logp)
        gradient = current_p * (torch.sum(current_p * (current_logp + 1.0)) - (current_logp + 1.0))
        gradient_norm = torch.sqrt(torch.sum(gradient ** 2))
        if float(gradient_norm) > 0:
            gradient = gradient / gradient_norm
            new_logits = current_logits - initial_alpha * gradient
            new_entropy = _entropy(new_logits)
            if float(new_entropy) < float(current_entropy):
                current_logits = new_logits
                if float(new_entropy) < float(best_entropy):
                    best_logits = new_logits
                    best_entropy = new_entropy
            else:
                initial_alpha *= 0.65
                if initial_alpha < 1e-4:
                    break
        else:
            break
    return best_logits, best_entropy


def get_input_ids(prompt_cache, model, ids_with_gen, ids_no_gen, model_key, use_cache: bool = True):
    """infer_opt.py:10-28: ``(tokens to feed, cache_hit, per-layer KV cache)``.  Without a prompt cache (or with
    ``use_cache=False``) a fresh cache and the whole prompt; otherwise whatever ``PromptCache.get_prompt_cache`` says is
    new.  Deviation (reference bug not reproduced): when a hit leaves NOTHING new to process the reference feeds the whole
    prompt again on top of the cache that already holds it; here the cache is trimmed by one position and the last prompt
    token is fed again, which yields the same logits the first processing did."""
    if not use_cache or prompt_cache is None:
        return list(ids_with_gen), False, make_prompt_cache(model)
    new_tokens, cache, cache_hit = prompt_cache.get_prompt_cache(model, ids_with_gen, ids_no_gen, model_key)
    if cache_hit:
        print(f"Use input cache 🐈 Processing {len(new_tokens)} tokens instead of {len(ids_with_gen)}")
        if len(new_tokens) == 0:
            for c in cache:
                if getattr(c, "offset", 0) > 0 and c.is_trimmable():
                    c.trim(1)
            new_tokens = list(ids_with_gen[-1:])
        return list(new_tokens), True, cache
    print("No cache benefit - processing all tokens")
    return list(ids_with_gen), False, cache


def _eminf_prefill(model, prompt_cache, input_ids, input_ids_no_gen, use_cache, max_kv_size):
    if max_kv_size is not None:
        raise NotImplementedError("RotatingKVCache (max_kv_size) is outside the QuantizedLinear hot path")
    model_key = getattr(model, "model_key", id(model))
    ids, _, cache = get_input_ids(prompt_cache, model, list(input_ids), list(input_ids_no_gen) if input_ids_no_gen is not None
                                  else list(input_ids), model_key, use_cache)
    dev = next(model.parameters()).device
    out = model(torch.tensor([ids], dtype=torch.long, device=dev), cache=cache, hidden_states=False)
    return out[:, -1, :].to(torch.float32), cache, dev


def _eminf_sample(logits, alpha, num_steps, threshold, rng):
    best_logits, _ = eminf_optimize(logits.reshape(-1), alpha=alpha, num_steps=num_steps, threshold=threshold)
    probs = torch.softmax(best_logits, dim=-1).double().cpu().numpy().ravel()
    probs = probs / probs.sum()
    tok = int(rng.choice(len(probs), p=probs)) if rng is not None else int(np.random.choice(len(probs), p=probs))
    return tok, best_logits


@torch.no_grad()
def eminf_generate(model, tokenizer, input_ids: List[int], input_ids_no_gen: Optional[List[int]] = None,
                   max_tokens: int = 256, num_steps: Optional[int] = None, alpha: float = 0.65, threshold: float = 0.05,
                   max_kv_size=None, prompt_cache=None, use_cache: bool = True, rng: Optional[np.random.Generator] = None,
                   verbose: bool = False) -> List[int]:
    """infer_opt.py:79-142.  ``rng`` (numpy Generator) makes the sampling step reproducible; the reference draws from
    the global numpy RNG (``np.random.choice``), which is what happens when ``rng`` is None."""
    logits, cache, dev = _eminf_prefill(model, prompt_cache, input_ids, input_ids_no_gen, use_cache, max_kv_size)
    generated: List[int] = []
    if not isinstance(tokenizer, TokenizerWrapper):
        tokenizer = TokenizerWrapper(tokenizer)
    detok = tokenizer.detokenizer
    detok.reset()
    fast = getattr(model, "decode_step_checked", None) or getattr(model, "decode_step", None)   # (hand-off faults are replayed: decode.py)
    for _ in range(max_tokens):
        tok, _ = _eminf_sample(logits, alpha, num_steps, threshold, rng)
        generated.append(tok)
        if tok == getattr(tokenizer, "eos_token_id", None) or tok in tokenizer.eos_token_ids:
            break
        detok.add_token(tok)
        seg = detok.last_segment
        if verbose and seg:
            print(seg, flush=True, end="")
        y = torch.tensor([tok], dtype=torch.long, device=dev)
        nxt = fast(y, cache) if fast is not None else model(y[None], cache=cache)[:, -1, :]
        logits = nxt.to(torch.float32)
    detok.finalize()
    return generated


@torch.no_grad()
def eminf_generate_step(model, tokenizer, input_ids: List[int], input_ids_no_gen: Optional[List[int]] = None,
                        max_tokens: int = 256, num_steps: Optional[int] = None, alpha: float = 0.65, threshold: float = 0.05,
                        max_kv_size=None, prompt_cache=None, use_cache: bool = True,
                        rng: Optional[np.random.Generator] = None):
    """infer_opt.py:168-245: the streaming form -- yields ``(token, logprob of the token under the optimised logits, None)``
    per generated token (``hidden_states`` is always None there too), stops BEFORE yielding an EOS token."""
    logits, cache, dev = _eminf_prefill(model, prompt_cache, input_ids, input_ids_no_gen, use_cache, max_kv_size)
    if not isinstance(tokenizer, TokenizerWrapper):
        tokenizer = TokenizerWrapper(tokenizer)
    fast = getattr(model, "decode_step_checked", None) or getattr(model, "decode_step", None)   # (hand-off faults are replayed: decode.py)
    for _ in range(max_tokens):
        tok, best_logits = _eminf_sample(logits, alpha, num_steps, threshold, rng)
        if tok == getattr(tokenizer, "eos_token_id", None) or tok in tokenizer.eos_token_ids:
            break
        logprobs = best_logits - torch.logsumexp(best_logits, dim=-1, keepdim=True)
        yield tok, logprobs[tok], None
        y = torch.tensor([tok], dtype=torch.long, device=dev)
        nxt = fast(y, cache) if fast is not None else model(y[None], cache=cache)[:, -1, :]
        logits = nxt.to(torch.float32)


def _chat_ids(tokenizer, messages):
    with_gen = tokenizer.apply_chat_template(messages, add_generation_prompt=True, enable_thinking=False)
    no_gen = tokenizer.apply_chat_template(messages, add_generation_prompt=False, enable_thinking=False)
    return with_gen, no_gen


def generate_response(model, tokenizer, messages, model_name, max_tokens, prompt_cache=None, use_cache: bool = True) -> str:
    """infer_opt.py:144-166: chat template (with / without the generation prompt), EMINF decode, decoded text."""
    with_gen, no_gen = _chat_ids(tokenizer, messages)
    ids = eminf_generate(model, tokenizer, with_gen, no_gen, max_tokens=max_tokens, prompt_cache=prompt_cache, use_cache=use_cache)
    return tokenizer.decode(np.array(ids), skip_special_tokens=True).strip()


def eminf_generate_response_stream(model, tokenizer, messages, model_name, max_tokens, prompt_cache=None, use_cache: bool = True):
    """infer_opt.py:247-284."""
    with_gen, no_gen = _chat_ids(tokenizer, messages)
    yield from eminf_generate_step(model, tokenizer, with_gen, no_gen, max_tokens=max_tokens, prompt_cache=prompt_cache,
                                   use_cache=use_cache)
