"""Qwen3 (dense) host graph: GQA attention with per-head RMSNorm on q and k, SwiGLU MLP (gbx_lm/models/qqwen3.py).

The attention is the one of the MoE variant (``qqwen3_moe.Attention``: q_norm / k_norm over head_dim before RoPE,
explicit ``head_dim``), the feed-forward the plain three-projection MLP.  Same class names / parameter paths as the
reference (``model.layers.<i>.self_attn.{q,k}_norm.weight`` ...); tied embeddings use ``embed_tokens`` as the head and
``sanitize`` drops ``lm_head.weight`` then (qqwen3.py:160-186).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn.functional as F
from torch import nn

from .base import BaseModelArgs, create_attention_mask
from .qllama import RMSNorm
from .qqwen3_moe import MLP, Attention as _QKNormAttention


@dataclass
class ModelArgs(BaseModelArgs):
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    rms_norm_eps: float
    vocab_size: int
    num_key_value_heads: int
    max_position_embeddings: int
    rope_theta: float
    head_dim: int
    tie_word_embeddings: bool
    rope_scaling: Optional[Dict[str, Union[float, str]]] = None


class Attention(_QKNormAttention):
    def __init__(self, args: ModelArgs):
        super().__init__(args, layer_idx=0)


class TransformerBlock(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.args = args
        self.hidden_size = args.hidden_size
        self.num_attention_heads = args.num_attention_heads
        self.self_attn = Attention(args)
        self.mlp = MLP(args.hidden_size, args.intermediate_size)
        self.input_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        h = x + self.self_attn(self.input_layernorm(x), mask, cache)
        return h + self.mlp(self.post_attention_layernorm(h))


class Qwen3Model(nn.Module):
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
        self.model = Qwen3Model(args)
        if not args.tie_word_embeddings:
            self.lm_head = nn.Linear(args.hidden_size, args.vocab_size, bias=False)
            self.lm_head.weight.requires_grad_(False)

    def head_weight(self) -> torch.Tensor:
        return self.model.embed_tokens.weight if self.args.tie_word_embeddings else self.lm_head.weight

    def forward(self, inputs, mask=None, cache=None, hidden_states: bool = False):
        out = self.model(inputs, mask, cache)
        logits = F.linear(out, self.head_weight())
        return (logits, out) if hidden_states else logits

    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if self.args.tie_word_embeddings:
            weights.pop("lm_head.weight", None)
        return weights

    def prefill(self, inputs, cache) -> None:
        """Fill the KV cache for a prompt chunk whose logits nobody reads (generate_step's chunk loop, gbx_lm/utils.py:312-319:
        the reference never evaluates those logits): the decoder body only, no output head."""
        self.model(inputs, None, cache)

    def last_logits(self, inputs, cache) -> torch.Tensor:
        """``self(inputs, cache=cache)[:, -1, :]`` (gbx_lm/utils.py:291-294) with the output head applied to the last
        position only -- at a 4096-token prompt the full head is 2 x 4096 x vocab x hidden flops of discarded work."""
        hidden = self.model(inputs, None, cache)
        return F.linear(hidden[:, -1:, :], self.head_weight())[:, -1, :]

    def decode_step(self, y: torch.Tensor, prompt_cache) -> torch.Tensor:
        """Single-token fast path: the hand-fused five-launch block of the Llama graph when the checkpoint qualifies (the
        per-head q / k norms run inside the attention launch), else this model's own forward captured once and replayed
        (gbx_lm_amd/decode.py:FusedDecodeRunner / GraphDecodeRunner)."""
        from ..decode import decode_step, fused_eligible, graph_decode_step
        ok = getattr(self, "_fused_ok", None)
        if ok is None:
            ok = self._fused_ok = bool(next(self.parameters()).is_cuda and fused_eligible(self))
        c0 = prompt_cache[0]
        if ok and getattr(c0, "keys", None) is not None and c0.keys.shape[0] == 1:
            return decode_step(self, y, prompt_cache)
        return graph_decode_step(self, y, prompt_cache)

    @property
    def layers(self):
        return self.model.layers
