"""Tokenizer plumbing for ``load()`` / ``stream_generate()`` (role of gbx_lm/tokenizer_utils.py:252-371).

Out of the optimisation scope: a thin wrapper that exposes ``eos_token_ids`` as a set
(tokenizer_utils.py:366-371) and a naive streaming detokenizer (re-decodes the running suffix).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Iterable, Optional


class NaiveStreamingDetokenizer:
    def __init__(self, tokenizer):
        self._tokenizer = tokenizer
        self.reset()

    def reset(self):
        self.tokens = []
        self._text = ""
        self._emitted = 0

    def add_token(self, token: int):
        self.tokens.append(int(token))

    def finalize(self):
        pass

    @property
    def text(self) -> str:
        self._text = self._tokenizer.decode(self.tokens) if self.tokens else ""
        return self._text

    @property
    def last_segment(self) -> str:
        text = self.text
        # hold back a trailing replacement char (incomplete multi-byte sequence)
        if text.endswith("�"):
            return ""
        seg = text[self._emitted:]
        self._emitted = len(text)
        return seg


class TokenizerWrapper:
    """Forwards to the wrapped HF tokenizer; adds ``detokenizer`` and ``eos_token_ids``."""

    def __init__(self, tokenizer, detokenizer_class=NaiveStreamingDetokenizer, eos_token_ids: Optional[Iterable[int]] = None):
        self._tokenizer = tokenizer
        self._detokenizer = detokenizer_class(tokenizer)
        if eos_token_ids is not None:
            self._eos_token_ids = set(int(t) for t in eos_token_ids)
        else:
            eid = getattr(tokenizer, "eos_token_id", None)
            self._eos_token_ids = {int(eid)} if eid is not None else set()

    def add_eos_token(self, token):
        token_id = None
        try:
            token_id = int(token)
        except (TypeError, ValueError):
            token_id = self._tokenizer.convert_tokens_to_ids(token)
        if token_id is None:
            raise ValueError(f"'{token}' is not a token for this tokenizer")
        self._eos_token_ids.add(token_id)

    @property
    def eos_token_ids(self):
        return self._eos_token_ids

    @property
    def detokenizer(self):
        return self._detokenizer

    def __getattr__(self, attr):
        if attr.startswith("_"):
            raise AttributeError(attr)
        return getattr(self._tokenizer, attr)


def load_tokenizer(model_path, tokenizer_config_extra={}, eos_token_ids=None) -> TokenizerWrapper:
    """tokenizer_utils.py:344-371: AutoTokenizer from the model directory; ``eos_token_ids`` may be int or list."""
    from transformers import AutoTokenizer, PreTrainedTokenizerFast

    model_path = Path(model_path)
    if isinstance(eos_token_ids, int):
        eos_token_ids = [eos_token_ids]
    try:
        tok = AutoTokenizer.from_pretrained(str(model_path), **tokenizer_config_extra)
    except Exception:
        tj = model_path / "tokenizer.json"
        if not tj.exists():
            raise
        cfg = {}
        tc = model_path / "tokenizer_config.json"
        if tc.exists():
            cfg = {k: v for k, v in json.loads(tc.read_text()).items() if k.endswith("_token")}
        tok = PreTrainedTokenizerFast(tokenizer_file=str(tj), **cfg)
    return TokenizerWrapper(tok, eos_token_ids=eos_token_ids)
