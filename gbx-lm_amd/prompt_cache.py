"""Conversation prompt-prefix cache (role of gbx_lm/prompt_cache.py:4-158): keep the KV cache of the turns already
processed and feed the model only the tokens a new request adds.

Same surface as the reference class -- ``PromptCache(quantize, qbit, q_group_size)``, ``cache_system_prompt``,
``get_prompt_cache(model, tokens_with_gen, tokens_no_gen, model_key) -> (tokens_to_process, cache, cache_hit)``,
``update_after_step`` -- and the same four decisions (prompt_cache.py:84-147):

1. no cache yet, or another model                      -> fresh cache, process everything, miss;
2. a pre-cached system prompt the request does not start with -> fresh cache, miss;
3. the request diverges from the cached conversation   -> if the shared prefix still covers the pre-cached system
   prompt: keep the cache and process the part after the shared prefix, hit; else fresh cache, miss;
4. the request extends the cached conversation         -> process only the new tokens (+ generation suffix), hit.

Two token sequences describe a request: ``tokens_no_gen`` (the conversation, what is compared and remembered) and
``tokens_with_gen`` (the same plus the generation prompt, what the model must see).

Deviation (a reference bug that is not reproduced): in case 3 the reference keeps the cache as it is, so the
KV entries of the abandoned branch stay in front of the new tokens; here the per-layer caches are trimmed back to the
shared prefix first (``KVCache.trim``), which is what the returned ``tokens_to_process`` assumes.
KV-cache quantisation (``quantize=True``) is outside the QuantizedLinear hot path and raises NotImplementedError.
"""
from __future__ import annotations

from typing import Any, List, Optional, Sequence, Tuple

import torch

from .models.cache import make_prompt_cache


class PromptCache:
    def __init__(self, quantize: bool = False, qbit=None, q_group_size=None):
        if quantize:
            raise NotImplementedError("KV-cache quantisation is outside the QuantizedLinear hot path")
        self.cache: Optional[List[Any]] = None
        self.tokens_no_gen: List[int] = []       # conversation WITHOUT generation prompt: what requests are compared with
        self.model_key = None
        self.system_cached = False
        self.system_tokens: List[int] = []
        self.quantize, self.qbit, self.q_group_size = quantize, qbit, q_group_size

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _common_prefix(a: Sequence[int], b: Sequence[int]) -> int:
        n = 0
        for x, y in zip(a, b):
            if x != y:
                break
            n += 1
        return n

    def _new_cache(self, model):
        return make_prompt_cache(model)

    def _restart(self, model, tokens_no_gen, model_key=None):
        self.cache = self._new_cache(model)
        if model_key is not None:
            self.model_key = model_key
        self.tokens_no_gen = list(tokens_no_gen)
        self.system_cached = False

    @staticmethod
    def _suffix(tokens_with_gen, tokens_no_gen) -> list:
        n = len(tokens_with_gen) - len(tokens_no_gen)
        return list(tokens_with_gen[-n:]) if n > 0 else []

    def _trim_to(self, n_tokens: int) -> None:
        for c in self.cache or []:
            extra = getattr(c, "offset", 0) - n_tokens
            if extra > 0 and c.is_trimmable():
                c.trim(extra)

    # ------------------------------------------------------------------ reference surface
    def cache_system_prompt(self, model, system_prompt: str, tokenizer) -> None:
        """Run the system prompt through the model once so that every later request starts from its KV entries."""
        model_key = getattr(model, "model_key", id(model))
        if self.system_cached and self.model_key == model_key:
            return
        self.system_tokens = list(tokenizer.apply_chat_template([{"role": "system", "content": system_prompt}],
                                                                 add_generation_prompt=False, enable_thinking=False))
        self.model_key = model_key
        cache = self._new_cache(model)
        dev = next(model.parameters()).device
        with torch.no_grad():
            model(torch.tensor([self.system_tokens], dtype=torch.long, device=dev), cache=cache)
        self.cache = cache
        self.system_cached = True
        self.tokens_no_gen = list(self.system_tokens)

    def get_prompt_cache(self, model, tokens_with_gen, tokens_no_gen, model_key) -> Tuple[list, List[Any], bool]:
        tokens_with_gen, tokens_no_gen = list(tokens_with_gen), list(tokens_no_gen)
        if self.cache is None or self.model_key != model_key:                       # case 1
            self._restart(model, tokens_no_gen, model_key)
            return tokens_with_gen, self.cache, False
        ns = len(self.system_tokens)
        if self.system_cached and tokens_no_gen[:ns] != self.system_tokens:         # case 2
            self._restart(model, tokens_no_gen)
            return tokens_with_gen, self.cache, False
        prefix = self._common_prefix(self.tokens_no_gen, tokens_no_gen)
        if prefix < len(self.tokens_no_gen):                                        # case 3: diverged
            if not (self.system_cached and prefix >= ns):
                self._restart(model, tokens_no_gen)
                return tokens_with_gen, self.cache, False
            self._trim_to(prefix)                                                   # (not in the reference: see module doc)
        todo = tokens_no_gen[prefix:] + self._suffix(tokens_with_gen, tokens_no_gen)    # cases 3 (kept) and 4
        self.tokens_no_gen = tokens_no_gen
        return todo, self.cache, True

    def update_after_step(self, response_text: str, tokenizer) -> None:
        """Remember the assistant turn the model just produced (as the chat template renders it)."""
        self.tokens_no_gen.extend(tokenizer.apply_chat_template([{"role": "assistant", "content": response_text}],
                                                                add_generation_prompt=False, enable_thinking=False))
        ns = len(self.system_tokens)
        self.system_cached = ns > 0 and self.tokens_no_gen[:ns] == self.system_tokens
