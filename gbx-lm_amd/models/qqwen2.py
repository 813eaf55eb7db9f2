"""Qwen2 host graph: the Llama block with biased q/k/v projections (gbx_lm/models/qqwen2.py).

Differences from qllama the reference makes (qqwen2.py:16-32,44-47,181-184): ``q_proj/k_proj/v_proj`` carry a bias,
``o_proj`` and the MLP do not; ``rope_theta`` defaults to 1e6 and ``max_position_embeddings`` to 32768;
``sanitize`` also drops ``lm_head.weight`` when the embeddings are tied.  Everything else -- and the decode hook, which
takes the generic graph-replayed runner because of the biases -- is the Llama graph.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Union

import torch
from torch import nn

from . import qllama
from .base import BaseModelArgs


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
    max_position_embeddings: int = 32768
    rope_theta: float = 1000000
    rope_traditional: bool = False
    rope_scaling: Optional[Dict[str, Union[float, str]]] = None
    tie_word_embeddings: bool = True


class Attention(qllama.Attention):
    def __init__(self, args: ModelArgs):
        super().__init__(args, qkv_bias=True, o_bias=False)


class TransformerBlock(qllama.TransformerBlock):
    def __init__(self, args: ModelArgs):
        super().__init__(args)
        self.self_attn = Attention(args)


class Qwen2Model(qllama.LlamaModel):
    def __init__(self, args: ModelArgs):
        super().__init__(args)
        self.layers = nn.ModuleList(TransformerBlock(args) for _ in range(args.num_hidden_layers))


class Model(qllama.Model):
    def __init__(self, args: ModelArgs):
        super().__init__(args)
        self.model = Qwen2Model(args)

    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if self.args.tie_word_embeddings:
            weights.pop("lm_head.weight", None)
        return {k: v for k, v in weights.items() if "self_attn.rotary_emb.inv_freq" not in k}
