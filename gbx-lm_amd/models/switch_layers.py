"""Expert-indexed (mixture-of-experts) linears around the GBA packed format.

PyTorch-ROCm counterpart of gbx_lm/models/switch_layers.py: ``QuantizedSwitchLinear`` (:26-92) keeps the reference's
constructor ``(input_dims, output_dims, num_experts, bias=True, group_size=64, bits=4)``, its parameters
``qweight [E, N, K*bits/32]`` / ``scales, zeros [E, N, K/g]`` / ``bias [E, N]`` and its call ``(x, indices,
sorted_indices=False)``; the arithmetic the reference delegates to ``mx.gather_qmm`` runs in the HIP library
(``ops.gather_quantized_matmul`` -> ``gbx_gather_qlinear_forward`` / per-expert ``gbx_qlinear_forward``).
``SwitchGLU`` (:157-203) routes every token through its top-k experts; sorting tokens by expert (:186-200) only
changes the order of evaluation in the reference and is subsumed by the grouped path here.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops


class QuantizedSwitchLinear(nn.Module):
    def __init__(self, input_dims: int, output_dims: int, num_experts: int, bias: bool = True,
                 group_size: int = 64, bits: int = 4):
        super().__init__()
        self.scale = math.sqrt(1 / input_dims)
        self.input_dims, self.output_dims = input_dims, output_dims
        self.group_size, self.bits, self.num_experts = group_size, bits, num_experts
        for nm in ("qweight", "scales", "zeros", "bias"):
            self.register_buffer(nm, None)
        self.init_params()
        if bias:
            self.bias = torch.zeros((num_experts, output_dims), dtype=torch.float16)

    def init_params(self, *_ignored) -> None:
        """Allocate the packed tensors for the current (bits, group_size) (the reference quantises a random matrix
        with mx.quantize, switch_layers.py:50-60; the values are overwritten by the checkpoint either way)."""
        e, n, k = self.num_experts, self.output_dims, self.input_dims
        self.qweight = torch.zeros((e, n, k // 32 * self.bits), dtype=torch.int32).view(torch.uint32)
        self.scales = torch.zeros((e, n, k // self.group_size), dtype=torch.float16)
        self.zeros = torch.zeros((e, n, k // self.group_size), dtype=torch.float16)

    def set_bias_and_weight(self) -> None:
        """GBA-native zero point -> MLX sign (switch_layers.py:62-71)."""
        self.zeros = -self.zeros

    def create_scales_zeros(self) -> None:       # double quantisation does not apply to the stacked experts
        return

    def extra_repr(self) -> str:
        return (f"input_dims={self.input_dims}, output_dims={self.output_dims}, num_experts={self.num_experts}, "
                f"bias={self.bias is not None}, group_size={self.group_size}, bits={self.bits}")

    def forward(self, x: torch.Tensor, indices: torch.Tensor, sorted_indices: bool = False) -> torch.Tensor:
        return ops.gather_quantized_matmul(x, self.qweight, scales=self.scales, biases=self.zeros, rhs_indices=indices,
                                           transpose=True, group_size=self.group_size, bits=self.bits,
                                           sorted_indices=sorted_indices, bias=self.bias)


class SwitchLinear(nn.Module):
    """Unquantised expert-indexed linear (switch_layers.py:95-141): weight [E, N, K]."""

    def __init__(self, input_dims: int, output_dims: int, num_experts: int, bias: bool = True):
        super().__init__()
        scale = math.sqrt(1 / input_dims)
        self.weight = nn.Parameter(torch.empty((num_experts, output_dims, input_dims)).uniform_(-scale, scale),
                                   requires_grad=False)
        self.bias = nn.Parameter(torch.zeros((num_experts, output_dims)), requires_grad=False) if bias else None

    @property
    def input_dims(self): return self.weight.shape[2]
    @property
    def output_dims(self): return self.weight.shape[1]
    @property
    def num_experts(self): return self.weight.shape[0]

    def forward(self, x: torch.Tensor, indices: torch.Tensor, sorted_indices: bool = False) -> torch.Tensor:
        w = self.weight[indices]                                   # [..., k, N, K]
        y = torch.matmul(x, w.transpose(-1, -2))                   # [..., k, 1, N]
        if self.bias is not None:
            y = y + self.bias[indices].unsqueeze(-2)
        return y


class SwiGLU(nn.Module):
    def forward(self, x: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
        if (gate.is_cuda and gate.dtype in (torch.bfloat16, torch.float16) and gate.shape == x.shape and gate.numel() % 8 == 0
                and gate.is_contiguous() and x.is_contiguous()):
            return ops.swiglu(gate, x)                                # one launch, same two roundings
        return F.silu(gate) * x


class SwitchGLU(nn.Module):
    """Top-k gated MLP over stacked experts: down(act(up(x), gate(x))) per (token, slot)  (switch_layers.py:157-203)."""

    def __init__(self, input_dims: int, hidden_dims: int, num_experts: int, activation: Optional[nn.Module] = None,
                 bias: bool = False, quant: bool = True):
        super().__init__()
        cls = QuantizedSwitchLinear if quant else SwitchLinear
        self.gate_proj = cls(input_dims, hidden_dims, num_experts, bias=bias)
        self.up_proj = cls(input_dims, hidden_dims, num_experts, bias=bias)
        self.down_proj = cls(hidden_dims, input_dims, num_experts, bias=bias)
        self.activation = activation if activation is not None else SwiGLU()

    def forward(self, x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        x = x.unsqueeze(-2).unsqueeze(-3)                          # [..., 1, 1, D]: one row shared by the k slots
        act = None
        if (isinstance(self.activation, SwiGLU) and isinstance(self.gate_proj, QuantizedSwitchLinear)
                and isinstance(self.up_proj, QuantizedSwitchLinear) and x.is_cuda):
            act = ops.gather_swiglu(x, self.gate_proj, self.up_proj, indices)    # one expert-indexed launch, or None
        if act is None:
            x_up = self.up_proj(x, indices)
            x_gate = self.gate_proj(x, indices)
            act = self.activation(x_up, x_gate)
        y = self.down_proj(act, indices)                            # [..., k, 1, D]
        return y.squeeze(-2)
