"""KV cache for the decode loop (plain ``KVCache`` of gbx_lm/models/cache.py:215-277; the rotating /
quantised / mamba caches and prompt-cache files are outside the hot path)."""
from __future__ import annotations

from typing import Any, List, Optional

import torch


def make_prompt_cache(model, max_kv_size: Optional[int] = None) -> List[Any]:
    """cache.py:11-36: defer to ``model.make_cache`` if present, else one KVCache per layer."""
    if hasattr(model, "make_cache"):
        return model.make_cache()
    if max_kv_size is not None:
        raise NotImplementedError("RotatingKVCache (max_kv_size) is outside the QuantizedLinear hot path")
    return [KVCache() for _ in range(len(model.layers))]


class KVCache:
    """Grows in ``step``-token blocks (cache.py:220-244); buffers are [B, n_kv, capacity, head_dim]."""

    def __init__(self, step: int = 256):
        self.keys = None
        self.values = None
        self.offset = 0
        self.step = step
        # graph-replayed decode (gbx_lm_amd/decode.py:GraphDecodeRunner): while ``static_active`` the write position is the
        # device-resident ``static_pos`` (int64 [1]) and attention runs over the whole buffer under a position mask, so one
        # captured launch sequence serves every step
        self.static_pos = None
        self.static_active = False

    @property
    def capacity(self) -> int:
        return 0 if self.keys is None else self.keys.shape[2]

    def reserve(self, n_new: int, like_k: torch.Tensor, like_v: torch.Tensor) -> None:
        """Make room for ``n_new`` more positions (same growth rule as update_and_fetch)."""
        prev = self.offset
        if self.keys is None or (prev + n_new) > self.keys.shape[2]:
            B, n_kv_heads, _, k_head_dim = like_k.shape
            v_head_dim = like_v.shape[3]
            n_steps = (self.step + n_new - 1) // self.step
            new_k = torch.zeros((B, n_kv_heads, n_steps * self.step, k_head_dim), dtype=like_k.dtype, device=like_k.device)
            new_v = torch.zeros((B, n_kv_heads, n_steps * self.step, v_head_dim), dtype=like_v.dtype, device=like_v.device)
            if self.keys is not None:
                if prev % self.step != 0:
                    self.keys = self.keys[..., :prev, :]
                    self.values = self.values[..., :prev, :]
                self.keys = torch.cat([self.keys, new_k], dim=2)
                self.values = torch.cat([self.values, new_v], dim=2)
            else:
                self.keys, self.values = new_k, new_v

    def update_and_fetch(self, keys: torch.Tensor, values: torch.Tensor):
        if self.static_active:                      # one new position at the device-resident index; shapes never change
            if (keys.is_cuda and keys.shape[2] == 1 and keys.dtype in (torch.bfloat16, torch.float16) and values.dtype == keys.dtype
                    and self.keys.is_contiguous() and self.values.is_contiguous()):
                from .. import ops                  # both caches in one launch (gbx_kv_append)
                ops.kv_append(self.keys, self.values, keys, values, self.static_pos)
            else:
                self.keys.index_copy_(2, self.static_pos, keys)
                self.values.index_copy_(2, self.static_pos, values)
            return self.keys, self.values
        prev = self.offset
        self.reserve(keys.shape[2], keys, values)
        self.offset += keys.shape[2]
        self.keys[..., prev:self.offset, :] = keys
        self.values[..., prev:self.offset, :] = values
        return self.keys[..., :self.offset, :], self.values[..., :self.offset, :]

    @property
    def state(self):
        if self.offset == self.keys.shape[2]:
            return self.keys, self.values
        return self.keys[..., :self.offset, :], self.values[..., :self.offset, :]

    @state.setter
    def state(self, v):
        self.keys, self.values = v
        self.offset = self.keys.shape[2]

    def is_trimmable(self):
        return True

    def trim(self, n):
        n = min(self.offset, n)
        self.offset -= n
        return n
