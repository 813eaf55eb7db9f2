"""Llama / Mistral host graph around the 7 QuantizedLinear call sites per block
(host side of gbx_lm/models/qllama.py:39-209).  Plain torch for the glue (RMSNorm, RoPE, SDPA, KV cache);
every projection goes through the HIP QuantizedLinear.  The fused batch-1 decode path lives in
``gbx_lm_amd.decode`` and must agree with this graph.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional, Union

import torch
from torch import nn

from .base import BaseModelArgs, create_attention_mask, scaled_dot_product_attention
from .quantized_linear_gba import QuantizedLinear
from .rope_utils import initialize_rope


@dataclass
class ModelArgs(BaseModelArgs):
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    rms_norm_eps: float
    vocab_size: int
    head_dim: Optional[int] = None
    max_position_embeddings: Optional[int] = None
    num_key_value_heads: Optional[int] = None
    attention_bias: bool = False
    mlp_bias: bool = False
    rope_theta: float = 10000
    rope_traditional: bool = False
    rope_scaling: Optional[Dict[str, Union[float, str]]] = None
    tie_word_embeddings: bool = True

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads


class RMSNorm(nn.Module):
    """``mx.nn.RMSNorm``: fp32 statistics, result in x.dtype."""

    def __init__(self, dims: int, eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dims), requires_grad=False)
        self.eps = eps

    def forward(self, x):
        xf = x.to(torch.float32)
        y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)
        return (y * self.weight.to(torch.float32)).to(x.dtype)


class Attention(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        dim = args.hidden_size
        self.n_heads = n_heads = args.num_attention_heads
        self.n_kv_heads = n_kv_heads = args.num_key_value_heads
        self.head_dim = head_dim = args.head_dim or args.hidden_size // n_heads
        self.scale = head_dim ** -0.5
        attention_bias = getattr(args, "attention_bias", False)
        self.q_proj = QuantizedLinear(dim, n_heads * head_dim, bias=attention_bias)
        self.k_proj = QuantizedLinear(dim, n_kv_heads * head_dim, bias=attention_bias)
        self.v_proj = QuantizedLinear(dim, n_kv_heads * head_dim, bias=attention_bias)
        self.o_proj = QuantizedLinear(n_heads * head_dim, dim, bias=attention_bias)
        self.rope = initialize_rope(self.head_dim, args.rope_theta, args.rope_traditional, args.rope_scaling,
                                    args.max_position_embeddings)

    def forward(self, x, mask=None, cache=None):
        B, L, D = x.shape
        queries, keys, values = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        queries = queries.reshape(B, L, self.n_heads, -1).transpose(1, 2)
        keys = keys.reshape(B, L, self.n_kv_heads, -1).transpose(1, 2)
        values = values.reshape(B, L, self.n_kv_heads, -1).transpose(1, 2)
        if cache is not None:
            queries = self.rope(queries, offset=cache.offset)
            keys = self.rope(keys, offset=cache.offset)
            keys, values = cache.update_and_fetch(keys, values)
        else:
            queries = self.rope(queries)
            keys = self.rope(keys)
        output = scaled_dot_product_attention(queries, keys, values, cache=cache, scale=self.scale, mask=mask)
        output = output.transpose(1, 2).reshape(B, L, -1)
        out = self.o_proj(output)
        if getattr(self, "tp_world", 1) > 1:       # K-split layer: partial sums meet here (gbx_lm_amd/tp.py)
            from ..tp import all_reduce_
            all_reduce_(out, self.tp_group)
        return out


class MLP(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        dim, hidden_dim = args.hidden_size, args.intermediate_size
        mlp_bias = getattr(args, "mlp_bias", False)
        self.gate_proj = QuantizedLinear(dim, hidden_dim, bias=mlp_bias)
        self.down_proj = QuantizedLinear(hidden_dim, dim, bias=mlp_bias)
        self.up_proj = QuantizedLinear(dim, hidden_dim, bias=mlp_bias)

    def forward(self, x):
        out = self.down_proj(nn.functional.silu(self.gate_proj(x)) * self.up_proj(x))
        if getattr(self, "tp_world", 1) > 1:
            from ..tp import all_reduce_
            all_reduce_(out, self.tp_group)
        return out


class TransformerBlock(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.num_attention_heads = args.num_attention_heads
        self.hidden_size = args.hidden_size
        self.self_attn = Attention(args)
        self.mlp = MLP(args)
        self.input_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.args = args

    def forward(self, x, mask=None, cache=None):
        r = self.self_attn(self.input_layernorm(x), mask, cache)
        h = x + r
        r = self.mlp(self.post_attention_layernorm(h))
        return h + r


class LlamaModel(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.args = args
        self.vocab_size = args.vocab_size
        self.num_hidden_layers = args.num_hidden_layers
        assert self.vocab_size > 0
        self.embed_tokens = nn.Embedding(args.vocab_size, args.hidden_size)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList([TransformerBlock(args=args) for _ in range(args.num_hidden_layers)])
        self.norm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

    def forward(self, inputs, mask=None, cache=None):
        h = self.embed_tokens(inputs)
        if mask is None:
            mask = create_attention_mask(h, cache)
        if cache is None:
            cache = [None] * len(self.layers)
        for layer, c in zip(self.layers, cache):
            h = layer(h, mask, cache=c)
        return self.norm(h)


class Model(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.args = args
        self.model_type = args.model_type
        self.model = LlamaModel(args)
        if not args.tie_word_embeddings:
            self.lm_head = nn.Linear(args.hidden_size, args.vocab_size, bias=False)
            self.lm_head.weight.requires_grad_(False)

    def head_weight(self) -> torch.Tensor:
        return self.model.embed_tokens.weight if self.args.tie_word_embeddings else self.lm_head.weight

    def forward(self, inputs, mask=None, cache=None, hidden_states: bool = False):
        out = self.model(inputs, mask, cache)
        logits = nn.functional.linear(out, self.head_weight())
        return (logits, out) if hidden_states else logits

    def decode_step(self, y, prompt_cache):
        """Single-token fast path (HIP-graph replay); see gbx_lm_amd/decode.py."""
        from ..decode import decode_step
        return decode_step(self, y, prompt_cache)

    def sanitize(self, weights):
        return {k: v for k, v in weights.items() if "self_attn.rotary_emb.inv_freq" not in k}

    @property
    def layers(self):
        return self.model.layers
