"""Llama / Mistral host graph around the seven QuantizedLinear call sites of a decoder block.

PyTorch-ROCm counterpart of gbx_lm/models/qllama.py (Attention :39-96, MLP :99-115, TransformerBlock :118-141,
LlamaModel :144-174, Model :177-209): same class names, constructor arguments, parameter paths
(``model.layers.<i>.self_attn.q_proj.qweight`` ...) and call signatures, so checkpoints and callers carry over.
Glue ops (RMSNorm, RoPE, SDPA, KV cache) are plain torch here; every projection is the HIP QuantizedLinear.
Single-token steps bypass this eager graph through ``Model.decode_step`` (fused HIP decode block, graph-replayed,
see ``gbx_lm_amd/decode.py``), which must agree with it numerically -- tests compare the two.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn.functional as F
from torch import nn

from .base import (BaseModelArgs, cache_offset, create_attention_mask, fused_decode_attention, rope_keys_and_update_cache,
                   scaled_dot_product_attention, swiglu)
from .quantized_linear_gba import QuantizedLinear
from .rope_utils import initialize_rope


@dataclass
class ModelArgs(BaseModelArgs):
    """config.json keys consumed by the Llama graph (unknown keys are dropped by ``from_dict``)."""
    # required
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    rms_norm_eps: float
    vocab_size: int
    # optional, with the reference's defaults (qllama.py:24-32)
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
        if self.num_key_value_heads is None:        # MHA unless the config says GQA
            self.num_key_value_heads = self.num_attention_heads


class RMSNorm(nn.Module):
    """``mx.nn.RMSNorm``: statistics in fp32, a single rounding back to the activation dtype."""

    def __init__(self, dims: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dims), requires_grad=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and self.weight.dtype == x.dtype:
            from .. import ops                       # one HIP launch (gbx_rmsnorm), same arithmetic as the lines below
            return ops.rmsnorm(x, self.weight, self.eps)
        x32 = x.float()
        inv = torch.rsqrt(x32.square().mean(dim=-1, keepdim=True) + self.eps)
        return (x32 * inv * self.weight.float()).to(x.dtype)


def _heads(t: torch.Tensor, n: int) -> torch.Tensor:
    """[B, L, n*hd] -> [B, n, L, hd]"""
    b, l, _ = t.shape
    return t.view(b, l, n, -1).permute(0, 2, 1, 3)


class Attention(nn.Module):
    """GQA self-attention; q/k/v/o are the four attention QuantizedLinear call sites."""

    def __init__(self, args, qkv_bias: Optional[bool] = None, o_bias: Optional[bool] = None):
        super().__init__()
        hidden = args.hidden_size
        self.n_heads = args.num_attention_heads
        self.n_kv_heads = args.num_key_value_heads
        self.head_dim = getattr(args, "head_dim", None) or hidden // self.n_heads
        self.scale = self.head_dim ** -0.5
        use_bias = bool(getattr(args, "attention_bias", False))       # Llama: one switch for all four (qllama.py:50-58)
        qkv_bias = use_bias if qkv_bias is None else qkv_bias         # families derived from this graph override
        o_bias = use_bias if o_bias is None else o_bias
        q_out, kv_out = self.n_heads * self.head_dim, self.n_kv_heads * self.head_dim
        self.q_proj = QuantizedLinear(hidden, q_out, bias=qkv_bias)
        self.k_proj = QuantizedLinear(hidden, kv_out, bias=qkv_bias)
        self.v_proj = QuantizedLinear(hidden, kv_out, bias=qkv_bias)
        self.o_proj = QuantizedLinear(q_out, hidden, bias=o_bias)
        self.rope = initialize_rope(self.head_dim, args.rope_theta, getattr(args, "rope_traditional", False),
                                    getattr(args, "rope_scaling", None), args.max_position_embeddings)
        self.tp_world, self.tp_group = 1, None      # set by gbx_lm_amd.tp.shard_llama

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        b, l, _ = x.shape
        q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        ctx = fused_decode_attention(q, k, v, cache, self.rope, self.scale, self.n_heads, self.n_kv_heads, self.head_dim)
        if ctx is not None:                             # graph-replayed decode: RoPE, cache append and attention in one launch
            out = self.o_proj(ctx)
            if self.tp_world > 1:
                from ..tp import all_reduce_
                all_reduce_(out, self.tp_group)
            return out
        q, k, v = _heads(q, self.n_heads), _heads(k, self.n_kv_heads), _heads(v, self.n_kv_heads)
        start = cache_offset(cache)
        q = self.rope(q, offset=start)
        k, v = rope_keys_and_update_cache(self.rope, k, v, cache, start)
        ctx = scaled_dot_product_attention(q, k, v, cache=cache, scale=self.scale, mask=mask)
        out = self.o_proj(ctx.permute(0, 2, 1, 3).reshape(b, l, -1))
        if self.tp_world > 1:                       # K-split layer: partial sums meet here (gbx_lm_amd/tp.py)
            from ..tp import all_reduce_
            all_reduce_(out, self.tp_group)
        return out


class MLP(nn.Module):
    """SwiGLU feed-forward: gate/up/down are the three MLP QuantizedLinear call sites."""

    def __init__(self, args: ModelArgs):
        super().__init__()
        use_bias = bool(getattr(args, "mlp_bias", False))
        self.gate_proj = QuantizedLinear(args.hidden_size, args.intermediate_size, bias=use_bias)
        self.down_proj = QuantizedLinear(args.intermediate_size, args.hidden_size, bias=use_bias)
        self.up_proj = QuantizedLinear(args.hidden_size, args.intermediate_size, bias=use_bias)
        self.tp_world, self.tp_group = 1, None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        act = None
        if x.is_cuda and x.numel() // x.shape[-1] >= 256 and type(self.gate_proj) is QuantizedLinear and type(self.up_proj) is QuantizedLinear:
            from .. import ops                       # long prompt (or a short one with resident images): both projections and the SwiGLU as one dense matrix-core GEMM
            act = ops.quantized_swiglu(x, self.gate_proj, self.up_proj)
        if act is None:
            act = swiglu(self.gate_proj(x), self.up_proj(x))
        out = self.down_proj(act)
        if self.tp_world > 1:
            from ..tp import all_reduce_
            all_reduce_(out, self.tp_group)
        return out


class TransformerBlock(nn.Module):
    """Pre-norm residual block: h = x + attn(norm(x)); out = h + mlp(norm(h))."""

    def __init__(self, args: ModelArgs):
        super().__init__()
        self.args = args
        self.hidden_size = args.hidden_size
        self.num_attention_heads = args.num_attention_heads
        self.self_attn = Attention(args)
        self.mlp = MLP(args)
        self.input_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[Any] = None) -> torch.Tensor:
        h = x + self.self_attn(self.input_layernorm(x), mask, cache)
        return h + self.mlp(self.post_attention_layernorm(h))


class LlamaModel(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        assert args.vocab_size > 0
        self.args = args
        self.vocab_size = args.vocab_size
        self.num_hidden_layers = args.num_hidden_layers
        self.embed_tokens = nn.Embedding(args.vocab_size, args.hidden_size)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList(TransformerBlock(args) for _ in range(args.num_hidden_layers))
        self.norm = RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

    def forward(self, inputs: torch.Tensor, mask: Optional[torch.Tensor] = None, cache: Optional[List[Any]] = None):
        h = self.embed_tokens(inputs)
        if mask is None:
            mask = create_attention_mask(h, cache)
        caches = cache if cache is not None else [None] * len(self.layers)
        if self._fused_residuals(h):
            return self._forward_fused_residuals(h, mask, caches)
        for block, c in zip(self.layers, caches):
            h = block(h, mask, cache=c)
        return self.norm(h)

    def _fused_residuals(self, h: torch.Tensor) -> bool:
        """prompt-sized CUDA calls of the plain Llama block: every residual add runs inside the following RMSNorm launch"""
        # (rows of 512 values or fewer: the stand-alone norm is the wave-per-row kernel, whose sum of squares is ordered
        # differently from gbx_add_rmsnorm's -- keep the op-by-op loop there so both routes stay bit-identical)
        return (h.is_cuda and h.shape[1] > 1 and h.dtype in (torch.bfloat16, torch.float16) and h.shape[-1] % 8 == 0
                and 512 < h.shape[-1] <= 8192 and self.norm.weight.dtype == h.dtype and len(self.layers) > 0
                and all(type(b) is TransformerBlock for b in self.layers))

    def _forward_fused_residuals(self, h, mask, caches):
        """Same graph as the loop above (TransformerBlock.forward: h = x + attn(norm(x)); out = h + mlp(norm(h))) with each
        ``a + b`` folded into the norm that consumes it (gbx_add_rmsnorm: the add rounded to the activation dtype in place, then
        the norm -- the op-by-op roundings), 2 launches per block instead of 4."""
        from .. import ops
        resid = h.contiguous()                      # the embedding output: owned here, updated in place
        delta = None
        for block, c in zip(self.layers, caches):
            ln = block.input_layernorm
            n = ln(resid) if delta is None else ops.add_rmsnorm(resid, delta, ln.weight, ln.eps)
            r = block.self_attn(n, mask, c)
            ln = block.post_attention_layernorm
            n = ops.add_rmsnorm(resid, r.contiguous(), ln.weight, ln.eps)
            delta = block.mlp(n).contiguous()
        return ops.add_rmsnorm(resid, delta, self.norm.weight, self.norm.eps)


class Model(nn.Module):
    """``Model(args)(inputs, mask=None, cache=None, hidden_states=False)`` like the reference (qllama.py:186-199)."""

    def __init__(self, args: ModelArgs):
        super().__init__()
        self.args = args
        self.model_type = args.model_type
        self.model = LlamaModel(args)
        if not args.tie_word_embeddings:
            self.lm_head = nn.Linear(args.hidden_size, args.vocab_size, bias=False)
            self.lm_head.weight.requires_grad_(False)

    def head_weight(self) -> torch.Tensor:
        """[vocab, hidden] matrix of the output head (tied: the embedding table, ``embed_tokens.as_linear``)."""
        shard = getattr(self, "_tp_head", None)         # tensor parallel: this rank's rows of the vocabulary (tp.shard_llama)
        if shard is not None:
            return shard
        return self.model.embed_tokens.weight if self.args.tie_word_embeddings else self.lm_head.weight

    def _full_logits(self, logits):
        if getattr(self, "_tp_head", None) is None:
            return logits
        from ..tp import gather_vocab
        return gather_vocab(logits, self)

    def forward(self, inputs, mask=None, cache=None, hidden_states: bool = False):
        hidden = self.model(inputs, mask, cache)
        logits = self._full_logits(F.linear(hidden, self.head_weight()))
        return (logits, hidden) if hidden_states else logits

    def prefill(self, inputs, cache) -> None:
        """Fill the KV cache for a prompt chunk whose logits nobody reads (generate_step's chunk loop, gbx_lm/utils.py:312-319:
        the reference never evaluates those logits): the decoder body only, no output head."""
        self.model(inputs, None, cache)

    def last_logits(self, inputs, cache) -> torch.Tensor:
        """``self(inputs, cache=cache)[:, -1, :]`` (gbx_lm/utils.py:291-294) with the output head applied to the last
        position only -- at a 4096-token prompt the full head is 2 x 4096 x vocab x hidden flops of discarded work."""
        hidden = self.model(inputs, None, cache)
        return self._full_logits(F.linear(hidden[:, -1:, :], self.head_weight())[:, -1, :])

    def prompt_step(self, y: torch.Tensor, prompt_cache):
        """Short fresh prompts: the prompt pass as one replayed HIP graph per length bucket (decode.PromptRunner); None = not applicable."""
        from ..decode import prompt_step
        return prompt_step(self, y, prompt_cache)

    def decode_step(self, y: torch.Tensor, prompt_cache) -> torch.Tensor:
        """Single-token fast path (fused HIP decode block, HIP-graph replay); see gbx_lm_amd/decode.py."""
        from ..decode import decode_step
        return decode_step(self, y, prompt_cache)

    def decode_step_logprobs(self, y: torch.Tensor, prompt_cache):
        """One default-sampler step of generate_step (token, logprobs, packed status) from one graph replay, or None (decode.py)."""
        from ..decode import decode_step_logprobs
        return decode_step_logprobs(self, y, prompt_cache)

    def decode_step_checked(self, y: torch.Tensor, prompt_cache) -> torch.Tensor:
        """``decode_step`` that checks the fused launches' status words behind the step (callers that synchronise per token: EMINF)."""
        from ..decode import decode_step_checked
        return decode_step_checked(self, y, prompt_cache)

    def decode_fault_words(self):
        """Device status words of the decode runner's fused launches; generate_step reads them next to every token (decode.py)."""
        from ..decode import decode_fault_words
        return decode_fault_words(self)

    def decode_fallback(self, code: int, prompt_cache=None, offset=None) -> None:
        """A fused decode launch timed out on a hand-off: this model decodes on the five-launch block from here on; the caches are
        rewound to ``offset`` and the rows the faulted steps appended are cleared (decode.py)."""
        from ..decode import decode_fallback
        decode_fallback(self, code, prompt_cache, offset)

    def sanitize(self, weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Drop precomputed rotary tables some exporters leave in the checkpoint (qllama.py:201-205)."""
        return {k: v for k, v in weights.items() if "self_attn.rotary_emb.inv_freq" not in k}

    @property
    def layers(self):
        return self.model.layers
