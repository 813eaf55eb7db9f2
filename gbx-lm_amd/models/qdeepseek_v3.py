"""DeepSeek-V3 host graph: multi-head latent attention (low-rank q and kv projections) and a group-limited
mixture of experts with shared experts, every projection over the GBA packed format
(gbx_lm/models/qdeepseek_v3.py: ModelArgs :17-47, YaRN :50-125, attention :128-253, MLP :256-273, router :276-335,
MoE :338-363, layer :366-392, model :395-456, Model + sanitize :459-508).

QuantizedLinear call sites per block: q_a_proj -> q_b_proj (or q_proj), kv_a_proj_with_mqa -> kv_b_proj, o_proj, then
either the dense gate/up/down or the routed experts (QuantizedSwitchLinear x3) plus the shared experts' gate/up/down.
The router itself is an unquantised matrix.  The reference's pipeline-parallel split of the layer list (MLX
distributed send / recv, :410-424, :441-452) has no counterpart here: tensor parallelism is the build's multi-GPU
scheme and covers the Llama graph only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from .base import BaseModelArgs, cache_offset, create_attention_mask, scaled_dot_product_attention, swiglu
from .qllama import RMSNorm
from .quantized_linear_gba import QuantizedLinear
from .rope_utils import RoPE, YarnRoPE, yarn_get_mscale
from .switch_layers import SwitchGLU


@dataclass
class ModelArgs(BaseModelArgs):
    model_type: str = "deepseek_v3"
    vocab_size: int = 102400
    hidden_size: int = 4096
    intermediate_size: int = 11008
    moe_intermediate_size: int = 1407
    num_hidden_layers: int = 30
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    n_shared_experts: Optional[int] = None
    n_routed_experts: Optional[int] = None
    routed_scaling_factor: float = 1.0
    kv_lora_rank: int = 512
    q_lora_rank: Optional[int] = 1536
    qk_rope_head_dim: int = 64
    v_head_dim: int = 128
    qk_nope_head_dim: int = 128
    topk_method: str = "noaux_tc"
    scoring_func: str = "sigmoid"
    norm_topk_prob: bool = True
    n_group: Optional[int] = None
    topk_group: Optional[int] = None
    num_experts_per_tok: Optional[int] = None
    moe_layer_freq: int = 1
    first_k_dense_replace: int = 0
    max_position_embeddings: int = 2048
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    rope_scaling: Optional[Dict] = None
    attention_bias: bool = False


class DeepseekV3Attention(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.config = config
        self.hidden_size, self.num_heads = config.hidden_size, config.num_attention_heads
        self.q_lora_rank, self.kv_lora_rank = config.q_lora_rank, config.kv_lora_rank
        self.qk_rope_head_dim, self.qk_nope_head_dim, self.v_head_dim = config.qk_rope_head_dim, config.qk_nope_head_dim, config.v_head_dim
        self.q_head_dim = self.qk_nope_head_dim + self.qk_rope_head_dim
        self.scale = self.q_head_dim ** -0.5
        if self.q_lora_rank is None:
            self.q_proj = QuantizedLinear(self.hidden_size, self.num_heads * self.q_head_dim, bias=False)
        else:
            self.q_a_proj = QuantizedLinear(self.hidden_size, self.q_lora_rank, bias=config.attention_bias)
            self.q_a_layernorm = RMSNorm(self.q_lora_rank)
            self.q_b_proj = QuantizedLinear(self.q_lora_rank, self.num_heads * self.q_head_dim, bias=False)
        self.kv_a_proj_with_mqa = QuantizedLinear(self.hidden_size, self.kv_lora_rank + self.qk_rope_head_dim,
                                                  bias=config.attention_bias)
        self.kv_a_layernorm = RMSNorm(self.kv_lora_rank)
        self.kv_b_proj = QuantizedLinear(self.kv_lora_rank, self.num_heads * (self.qk_nope_head_dim + self.v_head_dim), bias=False)
        self.o_proj = QuantizedLinear(self.num_heads * self.v_head_dim, self.hidden_size, bias=config.attention_bias)
        rs = config.rope_scaling
        if rs is not None and rs.get("mscale_all_dim", 0):
            m = yarn_get_mscale(rs["factor"], rs["mscale_all_dim"])
            self.scale = self.scale * m * m
        self.rope = self.build_rope(config)

    def build_rope(self, config: ModelArgs) -> nn.Module:
        rs = config.rope_scaling
        if rs is None:
            return RoPE(self.qk_rope_head_dim, traditional=True, base=config.rope_theta)
        kw = {k: rs[k] for k in ("original_max_position_embeddings", "beta_fast", "beta_slow", "mscale", "mscale_all_dim") if k in rs}
        return YarnRoPE(dim=self.qk_rope_head_dim, max_position_embeddings=config.max_position_embeddings,
                        scaling_factor=rs["factor"], base=config.rope_theta, **kw)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        b, l, _ = x.shape
        q = self.q_proj(x) if self.q_lora_rank is None else self.q_b_proj(self.q_a_layernorm(self.q_a_proj(x)))
        q = q.view(b, l, self.num_heads, self.q_head_dim).permute(0, 2, 1, 3)
        q_nope, q_pe = torch.split(q, [self.qk_nope_head_dim, self.qk_rope_head_dim], dim=-1)
        ckv, k_pe = torch.split(self.kv_a_proj_with_mqa(x), [self.kv_lora_rank, self.qk_rope_head_dim], dim=-1)
        k_pe = k_pe.reshape(b, l, 1, self.qk_rope_head_dim).permute(0, 2, 1, 3)
        kv = self.kv_b_proj(self.kv_a_layernorm(ckv)).view(b, l, self.num_heads, -1).permute(0, 2, 1, 3)
        k_nope, values = torch.split(kv, [self.qk_nope_head_dim, self.v_head_dim], dim=-1)
        start = cache_offset(cache)
        q_pe, k_pe = self.rope(q_pe, offset=start), self.rope(k_pe, offset=start)
        keys = torch.cat([k_nope, k_pe.expand(-1, self.num_heads, -1, -1)], dim=-1)
        if cache is not None:
            keys, values = cache.update_and_fetch(keys, values.contiguous())
        queries = torch.cat([q_nope, q_pe], dim=-1)
        out = scaled_dot_product_attention(queries, keys, values, cache=cache, scale=self.scale, mask=mask)
        return self.o_proj(out.permute(0, 2, 1, 3).reshape(b, l, -1))


class DeepseekV3MLP(nn.Module):
    def __init__(self, config: ModelArgs, hidden_size: Optional[int] = None, intermediate_size: Optional[int] = None):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size if hidden_size is None else hidden_size
        self.intermediate_size = config.intermediate_size if intermediate_size is None else intermediate_size
        self.gate_proj = QuantizedLinear(self.hidden_size, self.intermediate_size, bias=False)
        self.up_proj = QuantizedLinear(self.hidden_size, self.intermediate_size, bias=False)
        self.down_proj = QuantizedLinear(self.intermediate_size, self.hidden_size, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.down_proj(swiglu(self.gate_proj(x), self.up_proj(x)))


def group_expert_select(gates, e_score_correction_bias, top_k: int, n_group: int, topk_group: int,
                        routed_scaling_factor: float, norm_topk_prob: bool):
    """noaux_tc routing (qdeepseek_v3.py:276-307): sigmoid scores + correction bias, keep the ``topk_group`` groups with
    the largest sum of their two best experts, take the ``top_k`` best experts among them, weight them by the UNBIASED
    scores (normalised if asked) times ``routed_scaling_factor``."""
    orig = torch.sigmoid(gates.float())
    scores = (orig + e_score_correction_bias.float()).unflatten(-1, (n_group, -1))
    group_scores = scores.topk(2, dim=-1).values.sum(dim=-1, keepdim=True)            # [..., n_group, 1]
    drop = n_group - topk_group
    if drop > 0:
        drop_idx = torch.topk(-group_scores, drop, dim=-2).indices                       # the weakest groups
        scores = scores.scatter(-2, drop_idx.expand(*drop_idx.shape[:-1], scores.shape[-1]), 0.0)
    scores = scores.flatten(-2, -1)
    inds = torch.topk(scores, top_k, dim=-1).indices
    w = torch.gather(orig, -1, inds)
    if top_k > 1 and norm_topk_prob:
        w = w / w.sum(dim=-1, keepdim=True)
    return inds, w * routed_scaling_factor


class MoEGate(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        assert config.topk_method == "noaux_tc", "Unsupported topk method."
        self.config = config
        self.top_k, self.norm_topk_prob = config.num_experts_per_tok, config.norm_topk_prob
        self.n_routed_experts, self.routed_scaling_factor = config.n_routed_experts, config.routed_scaling_factor
        self.n_group, self.topk_group = config.n_group, config.topk_group
        self.weight = nn.Parameter(torch.zeros((self.n_routed_experts, config.hidden_size)), requires_grad=False)
        self.e_score_correction_bias = nn.Parameter(torch.zeros((self.n_routed_experts,)), requires_grad=False)

    def forward(self, x: torch.Tensor):
        return group_expert_select(F.linear(x, self.weight.to(x.dtype)), self.e_score_correction_bias, self.top_k,
                                   self.n_group, self.topk_group, self.routed_scaling_factor, self.norm_topk_prob)


class DeepseekV3MoE(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.config = config
        self.num_experts_per_tok = config.num_experts_per_tok
        self.switch_mlp = SwitchGLU(config.hidden_size, config.moe_intermediate_size, config.n_routed_experts)
        self.gate = MoEGate(config)
        if config.n_shared_experts is not None:
            self.shared_experts = DeepseekV3MLP(config, intermediate_size=config.moe_intermediate_size * config.n_shared_experts)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        inds, scores = self.gate(x)
        y = self.switch_mlp(x, inds)
        y = (y * scores.unsqueeze(-1).to(y.dtype)).sum(dim=-2).to(y.dtype)
        if self.config.n_shared_experts is not None:
            y = y + self.shared_experts(x)
        return y


class DeepseekV3DecoderLayer(nn.Module):
    def __init__(self, config: ModelArgs, layer_idx: int):
        super().__init__()
        self.self_attn = DeepseekV3Attention(config)
        moe = config.n_routed_experts is not None and layer_idx >= config.first_k_dense_replace \
            and layer_idx % config.moe_layer_freq == 0
        self.mlp = DeepseekV3MoE(config) if moe else DeepseekV3MLP(config)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        h = x + self.self_attn(self.input_layernorm(x), mask, cache)
        return h + self.mlp(self.post_attention_layernorm(h))


class DeepseekV3Model(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList(DeepseekV3DecoderLayer(config, i) for i in range(config.num_hidden_layers))
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, x: torch.Tensor, cache: Optional[List[Any]] = None, mask: Optional[torch.Tensor] = None):
        h = self.embed_tokens(x)
        if mask is None:
            mask = create_attention_mask(h, cache)
        caches = cache if cache is not None else [None] * len(self.layers)
        for layer, c in zip(self.layers, caches):
            h = layer(h, mask, c)
        return self.norm(h)


class Model(nn.Module):
    """``Model(args)(inputs, cache=None, mask=None, hidden_states=False)`` -- note the reference's argument order here."""

    def __init__(self, config: ModelArgs):
        super().__init__()
        self.args = config
        self.model_type = config.model_type
        self.model = DeepseekV3Model(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.lm_head.weight.requires_grad_(False)

    def forward(self, inputs, cache=None, mask=None, hidden_states: bool = False):
        out = self.model(inputs, cache, mask)
        logits = self.lm_head(out)
        return (logits, out) if hidden_states else logits

    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Stack ``mlp.experts.<e>.<proj>.<t>`` into ``mlp.switch_mlp.<proj>.<t>``, drop per-expert channel_scale / q_perm
        and the multi-token-prediction layer 61 (qdeepseek_v3.py:481-497)."""
        for l in range(self.args.num_hidden_layers):
            prefix = f"model.layers.{l}"
            for m in ("gate_proj", "down_proj", "up_proj"):
                for t in ("qweight", "scales", "zeros"):
                    if f"{prefix}.mlp.experts.0.{m}.{t}" in weights:
                        parts = [weights.pop(f"{prefix}.mlp.experts.{e}.{m}.{t}") for e in range(self.args.n_routed_experts)]
                        weights[f"{prefix}.mlp.switch_mlp.{m}.{t}"] = torch.stack(parts)
                if f"{prefix}.mlp.experts.0.{m}.channel_scale" in weights and f"{prefix}.mlp.experts.0.{m}.q_perm" in weights:
                    for e in range(self.args.n_routed_experts):
                        weights.pop(f"{prefix}.mlp.experts.{e}.{m}.channel_scale")
                        weights.pop(f"{prefix}.mlp.experts.{e}.{m}.q_perm")
        return {k: v for k, v in weights.items() if not k.startswith("model.layers.61")}

    def decode_step(self, y: torch.Tensor, prompt_cache) -> torch.Tensor:
        """Single-token fast path: this graph captured once and replayed (gbx_lm_amd/decode.py:GraphDecodeRunner)."""
        from ..decode import graph_decode_step
        return graph_decode_step(self, y, prompt_cache)

    @property
    def layers(self):
        return self.model.layers
