"""Phi-3 host graph: fused projections (``qkv_proj``, ``gate_up_proj``) over the GBA packed format, su / longrope RoPE
(gbx_lm/models/qphi3.py: ModelArgs :14-44, Attention :47-110, MLP :113-128, block :131-155, model :158-190, Model :193-211).

One QuantizedLinear produces q, k and v (split at n_heads*hd and n_heads*hd + n_kv*hd, :92-95) and one produces gate and
up (split in halves, :125-127), so a block is four QuantizedLinear calls instead of seven.  The strategy entries are
found by the reference's substring scan: ``qkv_proj`` matches ``v_proj`` first and falls through to ``qkv_proj`` on
KeyError, likewise ``gate_up_proj`` (quantized_linear_gba.py:256-272; covered by tests/test_host.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn.functional as F
from torch import nn

from .base import (BaseModelArgs, cache_offset, create_attention_mask, rope_keys_and_update_cache, scaled_dot_product_attention,
                   swiglu)
from .qllama import RMSNorm
from .quantized_linear_gba import QuantizedLinear
from .rope_utils import RoPE, SuScaledRotaryEmbedding


@dataclass
class ModelArgs(BaseModelArgs):
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    rms_norm_eps: float
    vocab_size: int
    num_key_value_heads: Optional[int] = None
    rope_theta: float = 10000
    rope_traditional: bool = False
    rope_scaling: Optional[Dict[str, Union[float, List[float]]]] = None
    max_position_embeddings: int = 131072
    original_max_position_embeddings: int = 4096

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        if self.rope_scaling:
            required = {"long_factor", "type"}
            if not all(k in self.rope_scaling for k in required):
                raise ValueError(f"rope_scaling must contain keys {required}")
            if self.rope_scaling["type"] not in ["longrope", "su", "linear"]:
                print("[WARNING] rope_scaling 'type' currently only supports 'linear', 'su', and 'longrope'; "
                      "setting rope scaling to false.")
                self.rope_scaling = None


class Attention(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        hidden = args.hidden_size
        self.n_heads, self.n_kv_heads = args.num_attention_heads, args.num_key_value_heads
        self.num_hidden_layers = args.num_hidden_layers
        self.head_dim = hidden // self.n_heads
        self.scale = self.head_dim ** -0.5
        self.q_size, self.kv_size = self.n_heads * self.head_dim, self.n_kv_heads * self.head_dim
        self.qkv_proj = QuantizedLinear(hidden, self.q_size + 2 * self.kv_size, bias=False)
        self.o_proj = QuantizedLinear(self.q_size, hidden, bias=False)
        self.rope = self.build_rope(args)

    def build_rope(self, args: ModelArgs) -> nn.Module:
        rs = args.rope_scaling
        if rs and rs["type"] in ("longrope", "su"):
            return SuScaledRotaryEmbedding(self.head_dim, base=args.rope_theta,
                                           max_position_embeddings=args.max_position_embeddings,
                                           original_max_position_embeddings=args.original_max_position_embeddings,
                                           short_factor=rs["short_factor"], long_factor=rs["long_factor"])
        scale = 1.0
        if rs and rs["type"] == "linear":
            assert isinstance(rs["factor"], float)
            scale = 1 / rs["factor"]
        return RoPE(self.head_dim, traditional=args.rope_traditional, base=args.rope_theta, scale=scale)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        b, l, _ = x.shape
        q, k, v = torch.split(self.qkv_proj(x), [self.q_size, self.kv_size, self.kv_size], dim=-1)
        q = q.reshape(b, l, self.n_heads, -1).permute(0, 2, 1, 3)
        k = k.reshape(b, l, self.n_kv_heads, -1).permute(0, 2, 1, 3)
        v = v.reshape(b, l, self.n_kv_heads, -1).permute(0, 2, 1, 3)
        start = cache_offset(cache)
        q = self.rope(q, offset=start)
        k, v = rope_keys_and_update_cache(self.rope, k, v, cache, start)
        ctx = scaled_dot_product_attention(q, k, v, cache=cache, scale=self.scale, mask=mask)
        return self.o_proj(ctx.permute(0, 2, 1, 3).reshape(b, l, -1))


class MLP(nn.Module):
    def __init__(self, dim: int, hidden_dim: int):
        super().__init__()
        self.gate_up_proj = QuantizedLinear(dim, 2 * hidden_dim, bias=False)
        self.down_proj = QuantizedLinear(hidden_dim, dim, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        gate, up = torch.chunk(self.gate_up_proj(x), 2, dim=-1)
        return self.down_proj(swiglu(gate, up))


class TransformerBlock(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.num_attention_heads, self.hidden_size = args.num_attention_heads, args.hidden_size
        self.args = args
        self.self_attn = Attention(args)
        self.mlp = MLP(args.hidden_size, args.intermediate_size)
        self.input_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        h = x + self.self_attn(self.input_layernorm(x), mask, cache)
        return h + self.mlp(self.post_attention_layernorm(h))


class Phi3Model(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        assert args.vocab_size > 0
        self.args = args
        self.vocab_size, self.num_hidden_layers = args.vocab_size, args.num_hidden_layers
        self.embed_tokens = nn.Embedding(args.vocab_size, args.hidden_size)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList(TransformerBlock(args) for _ in range(args.num_hidden_layers))
        self.norm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

    def forward(self, inputs: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[List[Any]] = None):
        h = self.embed_tokens(inputs)
        if mask is None:
            mask = create_attention_mask(h, cache)
        caches = cache if cache is not None else [None] * len(self.layers)
        for layer, c in zip(self.layers, caches):
            h = layer(h, mask, c)
        return self.norm(h)


class Model(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.args = args
        self.model_type = args.model_type
        self.model = Phi3Model(args)
        self.lm_head = nn.Linear(args.hidden_size, args.vocab_size, bias=False)
        self.lm_head.weight.requires_grad_(False)

    def forward(self, inputs, mask=None, cache=None, hidden_states: bool = False):
        out = self.model(inputs, mask, cache)
        logits = self.lm_head(out)
        return (logits, out) if hidden_states else logits

    def decode_step(self, y: torch.Tensor, prompt_cache) -> torch.Tensor:
        """Single-token fast path: this graph captured once and replayed (gbx_lm_amd/decode.py:GraphDecodeRunner)."""
        from ..decode import graph_decode_step
        return graph_decode_step(self, y, prompt_cache)

    @property
    def layers(self):
        return self.model.layers
