"""Qwen3-MoE host graph: the dense attention projections plus an expert-routed MLP over the GBA packed format.

PyTorch-ROCm counterpart of gbx_lm/models/qqwen3_moe.py (ModelArgs :15-35, Attention :38-98 with per-head q/k RMSNorm,
MLP :101-109, Qwen3MoeSparseMoeBlock :112-141, decoder layer :144-174, model :177-209, Model + sanitize :212-248): same
class names, parameter paths (``model.layers.<i>.mlp.switch_mlp.{gate,up,down}_proj.{qweight,scales,zeros}``,
``model.layers.<i>.mlp.gate.qweight`` ...) and call signatures.  Every projection is a HIP QuantizedLinear /
QuantizedSwitchLinear; routing (softmax, top-k, renormalisation, weighted sum) is plain torch.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from .base import (BaseModelArgs, cache_offset, create_attention_mask, fused_decode_attention, rope_keys_and_update_cache,
                   scaled_dot_product_attention, swiglu)
from .qllama import RMSNorm
from .quantized_linear_gba import QuantizedLinear
from .rope_utils import RoPE
from .switch_layers import SwitchGLU


@dataclass
class ModelArgs(BaseModelArgs):
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    num_experts_per_tok: int
    num_experts: int
    decoder_sparse_step: int
    mlp_only_layers: List[int]
    moe_intermediate_size: int
    rms_norm_eps: float
    vocab_size: int
    num_key_value_heads: int
    head_dim: int
    rope_theta: float
    tie_word_embeddings: bool
    max_position_embeddings: int
    norm_topk_prob: bool
    rope_scaling: Optional[Dict[str, Union[float, str]]] = None


class Attention(nn.Module):
    """GQA attention with RMSNorm over head_dim on q and k before RoPE (qqwen3_moe.py:56-57,76-81)."""

    def __init__(self, args: ModelArgs, layer_idx: int):
        super().__init__()
        hidden = args.hidden_size
        self.n_heads, self.n_kv_heads = args.num_attention_heads, args.num_key_value_heads
        self.head_dim = getattr(args, "head_dim", None) or hidden // self.n_heads
        self.scale = self.head_dim ** -0.5
        self.q_proj = QuantizedLinear(hidden, self.n_heads * self.head_dim, bias=False)
        self.k_proj = QuantizedLinear(hidden, self.n_kv_heads * self.head_dim, bias=False)
        self.v_proj = QuantizedLinear(hidden, self.n_kv_heads * self.head_dim, bias=False)
        self.o_proj = QuantizedLinear(self.n_heads * self.head_dim, hidden, bias=False)
        self.q_norm = RMSNorm(self.head_dim, eps=args.rms_norm_eps)
        self.k_norm = RMSNorm(self.head_dim, eps=args.rms_norm_eps)
        self.rope = RoPE(self.head_dim, traditional=False, base=args.rope_theta)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None,
                residual: Optional[torch.Tensor] = None, pre_norm: Optional[nn.Module] = None) -> torch.Tensor:
        """Decode fast path of the decoder layer: ``residual`` -> returns residual + attention(.); ``pre_norm`` -> x is the
        un-normalised stream and the layer norm runs in the prologue of the q|k|v launch."""
        b, l, _ = x.shape
        if pre_norm is not None and not x.is_cuda:
            x, pre_norm = pre_norm(x), None
        q, k, v = ops.quantized_matmul_multi(x, (self.q_proj, self.k_proj, self.v_proj), pre_norm) if x.is_cuda else \
            (self.q_proj(x), self.k_proj(x), self.v_proj(x))           # decode-sized x: one launch for the three
        ctx = fused_decode_attention(q, k, v, cache, self.rope, self.scale, self.n_heads, self.n_kv_heads, q.shape[-1] // self.n_heads,
                                     self.q_norm, self.k_norm)
        if ctx is not None:                             # graph-replayed decode: norms, RoPE, cache append and attention in one launch
            if residual is not None:                    # ... and the block's residual add in the o_proj launch
                out = ops.qlinear_residual(self.o_proj, ctx, residual)
                if out is not None:
                    return out
                return residual + self.o_proj(ctx)
            return self.o_proj(ctx)
        q = self.q_norm(q.view(b, l, self.n_heads, -1)).permute(0, 2, 1, 3)
        k = self.k_norm(k.view(b, l, self.n_kv_heads, -1)).permute(0, 2, 1, 3)
        v = v.view(b, l, self.n_kv_heads, -1).permute(0, 2, 1, 3)
        start = cache_offset(cache)
        q = self.rope(q, offset=start)
        k, v = rope_keys_and_update_cache(self.rope, k, v, cache, start)
        ctx = scaled_dot_product_attention(q, k, v, cache=cache, scale=self.scale, mask=mask)
        out = self.o_proj(ctx.permute(0, 2, 1, 3).reshape(b, l, -1))
        return out if residual is None else residual + out


class MLP(nn.Module):
    def __init__(self, dim: int, hidden_dim: int):
        super().__init__()
        self.gate_proj = QuantizedLinear(dim, hidden_dim, bias=False)
        self.down_proj = QuantizedLinear(hidden_dim, dim, bias=False)
        self.up_proj = QuantizedLinear(dim, hidden_dim, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.down_proj(swiglu(self.gate_proj(x), self.up_proj(x)))


def _device_routing(x: torch.Tensor, num_experts: int, top_k: int) -> bool:
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and num_experts <= 1024 and top_k <= 8
            and x.shape[-1] % 2 == 0 and x.numel() // x.shape[-1] <= 65535)


class Qwen3MoeSparseMoeBlock(nn.Module):
    """Router (a QuantizedLinear) -> softmax (fp32) -> top-k -> optional renormalisation -> SwitchGLU -> weighted sum
    (qqwen3_moe.py:112-141).  The reference picks the k largest with argpartition (unordered); topk here -- the set and
    the weighted sum are the same."""

    def __init__(self, args: ModelArgs):
        super().__init__()
        self.num_experts, self.top_k, self.norm_topk_prob = args.num_experts, args.num_experts_per_tok, args.norm_topk_prob
        self.gate = QuantizedLinear(args.hidden_size, args.num_experts, bias=False)
        self.switch_mlp = SwitchGLU(args.hidden_size, args.moe_intermediate_size, args.num_experts, quant=True)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``residual`` (decode fast path): returns residual + block(x), the add fused into the mixing launch."""
        if _device_routing(x, self.num_experts, self.top_k):
            # routing decisions and the weighted sum as two launches (gbx_moe_route / gbx_moe_combine)
            inds, scores = ops.moe_route(self.gate(x), self.top_k, 0, self.norm_topk_prob)
            return ops.moe_combine(self.switch_mlp(x, inds), scores, residual)
        if residual is not None:
            return residual + self.forward(x)
        gates = torch.softmax(self.gate(x).float(), dim=-1).to(x.dtype)
        scores, inds = torch.topk(gates, self.top_k, dim=-1)
        if self.norm_topk_prob:
            scores = scores / scores.sum(dim=-1, keepdim=True)
        y = self.switch_mlp(x, inds)
        return (y * scores.unsqueeze(-1)).sum(dim=-2)


class Qwen3MoeDecoderLayer(nn.Module):
    def __init__(self, args: ModelArgs, layer_idx: int):
        super().__init__()
        self.args = args
        self.hidden_size = args.hidden_size
        self.self_attn = Attention(args, layer_idx)
        self.input_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        sparse = (layer_idx not in args.mlp_only_layers) and args.num_experts > 0 \
            and (layer_idx + 1) % args.decoder_sparse_step == 0
        self.mlp = Qwen3MoeSparseMoeBlock(args) if sparse else MLP(args.hidden_size, args.intermediate_size)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        if getattr(cache, "static_active", False) and x.is_cuda and x.shape[1] == 1:
            # graph-replayed decode: both residual adds ride in the launch that produces the branch (o_proj GEMV / expert mixing)
            h = self.self_attn(x, mask, cache, residual=x, pre_norm=self.input_layernorm)
            if isinstance(self.mlp, Qwen3MoeSparseMoeBlock):
                return self.mlp(self.post_attention_layernorm(h), residual=h)
            return h + self.mlp(self.post_attention_layernorm(h))
        h = x + self.self_attn(self.input_layernorm(x), mask, cache)
        return h + self.mlp(self.post_attention_layernorm(h))


class Qwen3MoeModel(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        assert args.vocab_size > 0
        self.args = args
        self.vocab_size, self.num_hidden_layers = args.vocab_size, args.num_hidden_layers
        self.embed_tokens = nn.Embedding(args.vocab_size, args.hidden_size)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList(Qwen3MoeDecoderLayer(args, i) for i in range(args.num_hidden_layers))
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
        self.model = Qwen3MoeModel(args)
        self.lm_head = nn.Linear(args.hidden_size, args.vocab_size, bias=False)
        self.lm_head.weight.requires_grad_(False)

    def forward(self, inputs, mask=None, cache=None, hidden_states: bool = False):
        out = self.model(inputs, mask, cache)
        logits = self.lm_head(out)
        return (logits, out) if hidden_states else logits

    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Per-expert checkpoint tensors ``...mlp.experts.<e>.<proj>.<t>`` -> one stacked ``...mlp.switch_mlp.<proj>.<t>``
        of shape [E, ...]; per-expert channel_scale / q_perm are dropped (qqwen3_moe.py:227-245)."""
        for l in range(self.args.num_hidden_layers):
            prefix = f"model.layers.{l}"
            for n in ("up_proj", "down_proj", "gate_proj"):
                for t in ("qweight", "scales", "zeros"):
                    if f"{prefix}.mlp.experts.0.{n}.{t}" in weights:
                        parts = [weights.pop(f"{prefix}.mlp.experts.{e}.{n}.{t}") for e in range(self.args.num_experts)]
                        weights[f"{prefix}.mlp.switch_mlp.{n}.{t}"] = torch.stack(parts)
                if f"{prefix}.mlp.experts.0.{n}.channel_scale" in weights and f"{prefix}.mlp.experts.0.{n}.q_perm" in weights:
                    for e in range(self.args.num_experts):
                        weights.pop(f"{prefix}.mlp.experts.{e}.{n}.channel_scale")
                        weights.pop(f"{prefix}.mlp.experts.{e}.{n}.q_perm")
        return weights

    def decode_step(self, y: torch.Tensor, prompt_cache) -> torch.Tensor:
        """Single-token fast path: this graph captured once and replayed (gbx_lm_amd/decode.py:GraphDecodeRunner)."""
        from ..decode import graph_decode_step
        return graph_decode_step(self, y, prompt_cache)

    @property
    def layers(self):
        return self.model.layers
