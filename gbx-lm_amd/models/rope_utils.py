"""Rotary embeddings needed by the Llama graph: default / linear and Llama-3 frequency smoothing
(gbx_lm/models/rope_utils.py:70-122,197-257).  YaRN / su-rope belong to other families and raise.

``mx.fast.rope(x, dims, traditional=False, freqs=...)`` semantics restated in torch: angle of pair i at
position t is ``t * scale / freqs[i]``; non-traditional pairs element i with i + dims/2; computed in fp32,
returned in x.dtype.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn


def apply_rope(x: torch.Tensor, inv_freq: torch.Tensor, offset: int, scale: float, traditional: bool) -> torch.Tensor:
    """x [B, H, L, D]; rotates the first 2*len(inv_freq) features."""
    if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and inv_freq.is_cuda and inv_freq.dtype == torch.float32:
        from .. import ops                           # one HIP launch (gbx_rope) instead of ~10 elementwise kernels
        return ops.rope(x, inv_freq, offset, scale, traditional)
    L = x.shape[-2]
    dims = inv_freq.numel() * 2
    if isinstance(offset, torch.Tensor):            # device-resident position (graph-replayed decode)
        pos = (offset.to(torch.float32) + torch.arange(L, device=x.device, dtype=torch.float32)) * scale
    else:
        pos = (torch.arange(offset, offset + L, device=x.device, dtype=torch.float32) * scale)
    ang = pos[:, None] * inv_freq[None, :].to(x.device)
    cos, sin = torch.cos(ang), torch.sin(ang)
    xf = x[..., :dims].to(torch.float32)
    if traditional:
        x1, x2 = xf[..., 0::2], xf[..., 1::2]
        out = torch.stack([x1 * cos - x2 * sin, x1 * sin + x2 * cos], dim=-1).flatten(-2)
    else:
        x1, x2 = xf[..., : dims // 2], xf[..., dims // 2:]
        out = torch.cat([x1 * cos - x2 * sin, x1 * sin + x2 * cos], dim=-1)
    out = out.to(x.dtype)
    if dims < x.shape[-1]:
        out = torch.cat([out, x[..., dims:]], dim=-1)
    return out


class RoPE(nn.Module):
    def __init__(self, dims: int, traditional: bool = False, base: float = 10000, scale: float = 1.0):
        super().__init__()
        self.dims, self.traditional, self.scale = dims, traditional, scale
        inv = 1.0 / (base ** (torch.arange(0, dims, 2, dtype=torch.float32) / dims))
        self.register_buffer("inv_freq", inv, persistent=False)

    def forward(self, x, offset: int = 0):
        return apply_rope(x, self.inv_freq, offset, self.scale, self.traditional)


class Llama3RoPE(nn.Module):
    """rope_utils.py:70-122: low-frequency wavelengths scaled by ``factor``, medium ones smoothed."""

    def __init__(self, dims: int, max_position_embeddings: int = 2048, traditional: bool = False,
                 base: float = 10000, scaling_config: dict = None):
        super().__init__()
        self.dims, self.traditional = dims, traditional
        self.max_position_embeddings = max_position_embeddings
        factor = scaling_config["factor"]
        low_freq_factor = scaling_config.get("low_freq_factor", 1.0)
        high_freq_factor = scaling_config.get("high_freq_factor", 4.0)
        old_context_len = scaling_config.get("original_max_position_embeddings", 8192)
        low_freq_wavelen = old_context_len / low_freq_factor
        high_freq_wavelen = old_context_len / high_freq_factor
        freqs = base ** (torch.arange(0, dims, 2, dtype=torch.float32) / dims)
        wavelens = 2 * math.pi * freqs
        freqs = torch.where(wavelens > low_freq_wavelen, freqs * factor, freqs)
        is_medium = (wavelens > high_freq_wavelen) & (wavelens < low_freq_wavelen)
        smooth = (old_context_len / wavelens - low_freq_factor) / (high_freq_factor - low_freq_factor)
        smooth_freqs = freqs / ((1 - smooth) / factor + smooth)
        freqs = torch.where(is_medium, smooth_freqs, freqs)
        self.register_buffer("inv_freq", 1.0 / freqs, persistent=False)
        self.scale = 1.0

    def forward(self, x, offset: int = 0):
        return apply_rope(x, self.inv_freq, offset, 1.0, self.traditional)


class SuScaledRotaryEmbedding(nn.Module):
    """Phi-3 "su" / longrope embedding (gbx_lm/models/su_rope.py:11-65): per-frequency ``long_factor`` stretches the
    rotary periods and the input is scaled by sqrt(1 + ln(max / original) / ln(original)) (or ``long_mscale``) before
    the rotation.  ``short_factor`` / ``short_mscale`` are accepted and unused, like in the reference."""

    def __init__(self, dims: int, base: float = 10000.0, max_position_embeddings: int = 131072,
                 original_max_position_embeddings: int = 4096, short_factor=1.0, long_factor=1.0,
                 short_mscale: Optional[float] = None, long_mscale: Optional[float] = None):
        super().__init__()
        self.dims, self.traditional = dims, False
        periods = base ** (torch.arange(0, dims, 2, dtype=torch.float32) / dims)
        periods = torch.as_tensor(long_factor, dtype=torch.float32) * periods
        self.register_buffer("inv_freq", 1.0 / periods, persistent=False)
        self.original_max_position_embeddings = original_max_position_embeddings
        self.scale = long_mscale or math.sqrt(
            1 + math.log(max_position_embeddings / original_max_position_embeddings) / math.log(original_max_position_embeddings))

    def forward(self, x, offset: int = 0):
        return apply_rope((self.scale * x.float()).to(x.dtype), self.inv_freq, offset, 1.0, False)


def _yarn_correction_dim(num_rotations, dim, base, max_pos):
    return (dim * math.log(max_pos / (num_rotations * 2 * math.pi))) / (2 * math.log(base))


def yarn_get_mscale(scale: float = 1, mscale: float = 1) -> float:
    return 1.0 if scale <= 1 else 0.1 * mscale * math.log(scale) + 1.0


class YarnRoPE(nn.Module):
    """DeepSeek-V3 YaRN embedding (gbx_lm/models/qdeepseek_v3.py:49-125): interleaved ("traditional") rotation with
    per-frequency interpolation between the original and the ``scaling_factor``-stretched periods inside the
    [beta_fast, beta_slow] correction range, input scaled by mscale(factor, mscale) / mscale(factor, mscale_all_dim)."""

    def __init__(self, dim: int, max_position_embeddings: int = 2048, base: float = 10000, scaling_factor: float = 1.0,
                 original_max_position_embeddings: int = 4096, beta_fast: float = 32, beta_slow: float = 1,
                 mscale: float = 1, mscale_all_dim: float = 0):
        super().__init__()
        self.dims, self.traditional, self.scale = dim, True, 1.0
        self.mscale = yarn_get_mscale(scaling_factor, mscale) / yarn_get_mscale(scaling_factor, mscale_all_dim)
        freq_extra = base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim)
        freq_inter = scaling_factor * freq_extra
        low = max(math.floor(_yarn_correction_dim(beta_fast, dim, base, original_max_position_embeddings)), 0)
        high = min(math.ceil(_yarn_correction_dim(beta_slow, dim, base, original_max_position_embeddings)), dim - 1)
        hi = high + 0.001 if low == high else high                   # prevent a singular ramp
        ramp = torch.clip((torch.arange(dim // 2, dtype=torch.float32) - low) / (hi - low), 0, 1)
        freq_mask = 1.0 - ramp
        periods = (freq_inter * freq_extra) / (freq_inter * freq_mask + freq_extra * (1 - freq_mask))
        self.register_buffer("inv_freq", 1.0 / periods, persistent=False)

    def forward(self, x, offset: int = 0):
        if self.mscale != 1.0:
            x = (self.mscale * x.float()).to(x.dtype)
        return apply_rope(x, self.inv_freq, offset, 1.0, True)


def initialize_rope(dims, base, traditional, scaling_config: Optional[dict] = None,
                    max_position_embeddings: Optional[int] = None):
    """rope_utils.py:197-257 (dispatch on ``type`` / ``rope_type``)."""
    if scaling_config is not None:
        rope_type = scaling_config.get("type") or scaling_config.get("rope_type", "default")
    else:
        rope_type = "default"
    if rope_type in ["default", "linear"]:
        scale = 1 / scaling_config["factor"] if rope_type == "linear" else 1.0
        return RoPE(dims, traditional=traditional, base=base, scale=scale)
    if rope_type == "llama3":
        return Llama3RoPE(dims=dims, max_position_embeddings=max_position_embeddings, traditional=traditional,
                          base=base, scaling_config=scaling_config)
    if rope_type in ("yarn", "longrope"):
        raise NotImplementedError(f"RoPE type {rope_type} belongs to model families outside the Llama hot path")
    raise ValueError(f"Unsupported RoPE type {rope_type}")
