"""Mixtral host graph: Llama-style attention, every feed-forward a top-k mixture of stacked experts
(gbx_lm/models/qmixtral.py: ModelArgs :15-33, attention :36-96, MixtralSparseMoeBlock :99-123, layer :126-151,
model :154-187, Model + sanitize :190-224).

Graph deltas against Qwen3-MoE: the router is an UNQUANTISED ``nn.Linear`` named ``gate``; the top-k logits are
soft-maxed AFTER selection (:115-117); the block is called ``block_sparse_moe``; no q/k norm; the checkpoint names the
expert projections w1 / w2 / w3 (= gate / down / up).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

import torch
from torch import nn

from .. import ops
from . import qllama
from .base import BaseModelArgs, create_attention_mask
from .switch_layers import SwitchGLU


@dataclass
class ModelArgs(BaseModelArgs):
    model_type: str
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_experts_per_tok: int = 2
    num_key_value_heads: int = 8
    num_local_experts: int = 8
    rms_norm_eps: float = 1e-5
    rope_theta: float = 1e6
    rope_traditional: bool = False
    rope_scaling: Optional[Dict[str, Union[float, str]]] = None
    max_position_embeddings: Optional[int] = None

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads


class MixtralAttention(qllama.Attention):
    """Bias-free GQA attention, plain RoPE (the Llama attention with fixed switches)."""

    def __init__(self, args: ModelArgs):
        super().__init__(args, qkv_bias=False, o_bias=False)
        self.hidden_size, self.num_heads, self.num_key_value_heads = args.hidden_size, self.n_heads, self.n_kv_heads
        self.rope_theta = args.rope_theta


class MixtralSparseMoeBlock(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.hidden_dim, self.ffn_dim = args.hidden_size, args.intermediate_size
        self.num_experts, self.num_experts_per_tok = args.num_local_experts, args.num_experts_per_tok
        self.gate = nn.Linear(self.hidden_dim, self.num_experts, bias=False)
        self.gate.weight.requires_grad_(False)
        self.switch_mlp = SwitchGLU(self.hidden_dim, self.ffn_dim, self.num_experts)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        gates = self.gate(x)
        if (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and gates.dtype == x.dtype and self.num_experts <= 1024
                and self.num_experts_per_tok <= 8 and x.numel() // x.shape[-1] <= 65535):
            inds, scores = ops.moe_route(gates, self.num_experts_per_tok, 1, False)     # top-k of the logits, softmax over them
            return ops.moe_combine(self.switch_mlp(x, inds), scores)
        top, inds = torch.topk(gates, self.num_experts_per_tok, dim=-1)       # the reference's argpartition: same set
        scores = torch.softmax(top.float(), dim=-1).to(x.dtype)
        y = self.switch_mlp(x, inds)
        return (y * scores.unsqueeze(-1)).sum(dim=-2)


class MixtralDecoderLayer(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.hidden_size = args.hidden_size
        self.self_attn = MixtralAttention(args)
        self.block_sparse_moe = MixtralSparseMoeBlock(args)
        self.input_layernorm = qllama.RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.post_attention_layernorm = qllama.RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        h = x + self.self_attn(self.input_layernorm(x), mask, cache)
        return h + self.block_sparse_moe(self.post_attention_layernorm(h))


class MixtralModel(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.vocab_size, self.num_hidden_layers = args.vocab_size, args.num_hidden_layers
        self.embed_tokens = nn.Embedding(args.vocab_size, args.hidden_size)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList(MixtralDecoderLayer(args) for _ in range(args.num_hidden_layers))
        self.norm = qllama.RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

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
        self.model = MixtralModel(args)
        self.lm_head = nn.Linear(args.hidden_size, args.vocab_size, bias=False)
        self.lm_head.weight.requires_grad_(False)

    def forward(self, inputs, mask=None, cache=None, hidden_states: bool = False):
        out = self.model(inputs, mask, cache)
        logits = self.lm_head(out)
        return (logits, out) if hidden_states else logits

    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """``block_sparse_moe.experts.<e>.w{1,2,3}.<t>`` -> stacked ``block_sparse_moe.switch_mlp.{gate,down,up}_proj.<t>``
        (qmixtral.py:205-220; the tensors are GBA's qweight / scales / zeros here)."""
        for l in range(self.args.num_hidden_layers):
            prefix = f"model.layers.{l}.block_sparse_moe"
            for n, m in (("w1", "gate_proj"), ("w2", "down_proj"), ("w3", "up_proj")):
                for t in ("qweight", "scales", "zeros"):
                    if f"{prefix}.experts.0.{n}.{t}" in weights:
                        parts = [weights.pop(f"{prefix}.experts.{e}.{n}.{t}") for e in range(self.args.num_local_experts)]
                        weights[f"{prefix}.switch_mlp.{m}.{t}"] = torch.stack(parts)
        return weights

    def decode_step(self, y: torch.Tensor, prompt_cache) -> torch.Tensor:
        """Single-token fast path: this graph captured once and replayed (gbx_lm_amd/decode.py:GraphDecodeRunner)."""
        from ..decode import graph_decode_step
        return graph_decode_step(self, y, prompt_cache)

    @property
    def layers(self):
        return self.model.layers
