"""Model-graph glue shared by the dense families (host side of gbx_lm/models/base.py).

Only what the QuantizedLinear callers need: ``BaseModelArgs.from_dict`` (base.py:15-24), the additive
causal mask with cache offset (base.py:27-62) and the SDPA dispatch (base.py:101-122; the quantised-KV
branch :65-98 is out of scope and raises).
"""
from __future__ import annotations

import inspect
from dataclasses import dataclass
from typing import Any, Optional

import torch
import torch.nn.functional as F


@dataclass
class BaseModelArgs:
    @classmethod
    def from_dict(cls, params):
        return cls(**{k: v for k, v in params.items() if k in inspect.signature(cls).parameters})


def create_causal_mask(N: int, offset: int = 0, window_size: Optional[int] = None, device=None) -> torch.Tensor:
    rinds = torch.arange(offset + N, device=device)
    linds = torch.arange(offset, offset + N, device=device) if offset else rinds
    linds = linds[:, None]
    rinds = rinds[None]
    mask = linds < rinds
    if window_size is not None:
        mask = mask | (linds > rinds + window_size)
    return mask * -1e9


def create_attention_mask(h: torch.Tensor, cache: Optional[Any] = None) -> Optional[torch.Tensor]:
    T = h.shape[1]
    if T > 1:
        window_size = None
        offset = 0
        if cache is not None and cache[0] is not None:
            c = cache[0]
            if hasattr(c, "max_size"):
                offset = min(c.max_size, c.offset)
                window_size = c.max_size
            else:
                offset = c.offset
        mask = create_causal_mask(T, offset, window_size=window_size, device=h.device).to(h.dtype)
        if window_size is None:             # plain causal: the HIP prefill attention rebuilds it from the offset alone
            mask._gbx_causal_offset = offset
        return mask
    return None


def swiglu(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """``nn.silu(gate) * up`` (qllama.py:115): one HIP launch over both operands on the GPU (same two roundings to the
    activation dtype), the torch ops elsewhere."""
    if (gate.is_cuda and gate.dtype in (torch.bfloat16, torch.float16) and gate.shape == up.shape and up.dtype == gate.dtype
            and gate.numel() % 8 == 0 and gate.is_contiguous() and up.is_contiguous()):
        from .. import ops
        return ops.swiglu(gate, up)
    return F.silu(gate) * up


def fused_decode_attention(q, k, v, cache, rope, scale: float, n_heads: int, n_kv_heads: int, head_dim: int,
                           q_norm=None, k_norm=None):
    """One new token per sequence inside a graph-replayed decode step (static cache): optional per-head q / k RMSNorm, RoPE,
    cache append and grouped-query attention in ONE launch (gbx_attn_decode_normed) from the raw projections
    q [B, 1, Hq*D], k / v [B, 1, Hkv*D].  Returns the context [B, 1, Hq*D], or None when the layer is outside that launch
    (prefill, eager cache, partial or rescaled-amplitude RoPE, head_dim not 64 / 128) -- callers then run the op-by-op graph."""
    from .rope_utils import RoPE, Llama3RoPE
    if not (cache is not None and getattr(cache, "static_active", False) and q.is_cuda and q.shape[1] == 1
            and head_dim in (64, 128) and type(rope) in (RoPE, Llama3RoPE) and rope.inv_freq.numel() * 2 == head_dim
            and q.dtype in (torch.bfloat16, torch.float16) and cache.keys is not None and cache.keys.dtype == q.dtype
            and cache.keys.is_contiguous() and cache.values.is_contiguous() and cache.keys.shape[3] == head_dim
            and cache.values.shape[3] == head_dim and n_heads % n_kv_heads == 0):
        return None
    if (q_norm is None) != (k_norm is None) or (q_norm is not None and q_norm.weight.dtype != q.dtype):
        return None
    from .. import ops
    bsz = q.shape[0]
    q2, k2, v2 = (t.reshape(bsz, -1).contiguous() for t in (q, k, v))
    out = torch.empty_like(q2)
    ops.attn_decode_batched(q2, k2, v2, cache.keys, cache.values, rope.inv_freq, cache.static_pos, out, n_heads, n_kv_heads, head_dim,
                            getattr(rope, "scale", 1.0), scale, rope.traditional,
                            q_norm.weight if q_norm is not None else None, k_norm.weight if k_norm is not None else None,
                            q_norm.eps if q_norm is not None else 0.0)
    return out.reshape(bsz, 1, -1)


def rope_keys_and_update_cache(rope, k: torch.Tensor, v: torch.Tensor, cache, start):
    """``k = rope(k, offset=start); k, v = cache.update_and_fetch(k, v)`` (qllama.py:88-91).  A prompt chunk on the GPU with a plain
    full-width RoPE and a plain KVCache does both in ONE launch: the rotated keys go straight into the cache (gbx_rope_kv_write:
    the same arithmetic as the stand-alone RoPE launch, so the cache contents are bit-identical), no temporary, no copy launches."""
    from .cache import KVCache
    from .rope_utils import RoPE, Llama3RoPE
    L, D = k.shape[2], k.shape[3]
    if (type(cache) is KVCache and not cache.static_active and k.is_cuda and L >= 16 and type(rope) in (RoPE, Llama3RoPE)
            and isinstance(start, int) and rope.inv_freq.numel() * 2 == D and D % 16 == 0 and D <= 256 and v.shape[3] == D
            and k.dtype in (torch.bfloat16, torch.float16) and v.dtype == k.dtype and k.stride(3) == 1 and v.stride(3) == 1
            and all(t.stride(i) % 8 == 0 for t in (k, v) for i in (0, 1, 2)) and k.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0
            and rope.inv_freq.is_cuda and rope.inv_freq.dtype == torch.float32):
        prev = cache.offset
        cache.reserve(L, k, v)
        # (the kernel indexes the cache with k's batch and head counts: a reused cache of another geometry takes the generic path,
        # whose update_and_fetch raises the shape error)
        if (cache.keys.dtype == k.dtype and cache.values.dtype == k.dtype and cache.keys.is_contiguous() and cache.values.is_contiguous()
                and cache.keys.shape[3] == D and cache.values.shape[3] == D
                and tuple(cache.keys.shape[:2]) == tuple(k.shape[:2]) and tuple(cache.values.shape[:2]) == tuple(v.shape[:2])):
            from .. import ops
            ops.rope_kv_write(k, v, cache.keys, cache.values, rope.inv_freq, start, prev, getattr(rope, "scale", 1.0), rope.traditional)
            cache.offset = prev + L
            return cache.keys[..., :cache.offset, :], cache.values[..., :cache.offset, :]
        cache.offset = prev                              # fast path not taken: hand an untouched offset to the generic path
    k = rope(k, offset=start)
    if cache is not None:
        k, v = cache.update_and_fetch(k, v)
    return k, v


def cache_offset(cache):
    """Position of the next token: the cache's host-side ``offset``, or -- inside a graph-replayed decode step -- its
    device-resident ``static_pos`` (RoPE then reads the position from device memory)."""
    if cache is None:
        return 0
    return cache.static_pos if getattr(cache, "static_active", False) else cache.offset


def scaled_dot_product_attention(queries, keys, values, cache, scale: float, mask: Optional[torch.Tensor]):
    """[B, Hq, L, D] x [B, Hkv, S, D] -> [B, Hq, L, D]; GQA by head repetition like mx.fast.sdpa."""
    if getattr(cache, "bits", None) is not None:
        raise NotImplementedError("quantised KV cache is outside the QuantizedLinear hot path")
    if getattr(cache, "static_active", False):    # whole buffer, positions past the device-resident index masked out
        if (queries.is_cuda and queries.shape[2] == 1 and queries.shape[3] in (64, 128) and values.shape[3] == queries.shape[3]
                and queries.dtype in (torch.bfloat16, torch.float16) and keys.dtype == queries.dtype
                and keys.is_contiguous() and values.is_contiguous()):
            from .. import ops                     # one launch: K / V read once per kv group (gbx_attn_cached)
            return ops.attn_cached(queries, keys, values, cache.static_pos, scale)
        cap = keys.shape[2]
        live = torch.arange(cap, device=keys.device)[None, None, None, :] <= cache.static_pos
        mask = torch.where(live, 0.0, float("-inf")).to(queries.dtype)
    off = getattr(mask, "_gbx_causal_offset", None)
    if off is not None and queries.shape[2] > 1 and keys.shape[2] == off + queries.shape[2]:
        from .. import ops                         # one launch on the matrix cores, no mask tensor, no repeated K / V heads
        if ops.attn_prefill_supported(queries, keys, values):
            b, hq, l, d = queries.shape
            return ops.attn_prefill(queries, keys, values, off, scale).view(b, l, hq, d).permute(0, 2, 1, 3)
    n_q, n_kv = queries.shape[1], keys.shape[1]
    if n_q != n_kv:
        rep = n_q // n_kv
        keys = keys.repeat_interleave(rep, dim=1)
        values = values.repeat_interleave(rep, dim=1)
    return F.scaled_dot_product_attention(queries, keys, values, attn_mask=mask, scale=scale)
