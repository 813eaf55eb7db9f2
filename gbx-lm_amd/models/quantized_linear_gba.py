"""GBA low-bit ``QuantizedLinear`` on MI355X: torch module, HIP kernels underneath.

Host-side mirror of gbx_lm/models/quantized_linear_gba.py (reference): same constructor
signature ``(input_dims, output_dims, bias=False, group_size=64, bits=4)`` (:36-43), same
tensor names / shapes / dtypes (:60-117), same three loader hooks (:210-324).  The forward
(:179-207) is one call across the C-ABI (``gbx_qlinear_forward``) instead of
``mx.quantized_matmul``.  Tensors are registered as buffers: the reference freezes them
(:57-58,162-166), so there is no backward.

Deliberately NOT reproduced (reference bugs): ``_extra_repr``'s swapped names and
``32 // bits`` assert (:168-177, wrong for 3 bits); ``from_linear``'s undefined ``weight``
(:334-344).
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
from torch import nn

from .. import ops

# substring scan order of the reference (:261); order matters (``qkv_proj`` hits ``v_proj`` first)
_STRATEGY_KEYS = ['kv_a_proj_with_mqa', 'kv_b_proj', 'q_a_proj', 'q_b_proj', 'q_proj', 'k_proj', 'v_proj',
                  'o_proj', 'gate_proj', 'up_proj', 'down_proj', 'qkv_proj', 'gate_up_proj']

_DQ_NAMES = ("qstatistic", "qzeros_zeros", "qzeros_scales", "qscales_zeros", "qscales_scales")


def _is_gba_linear(m) -> bool:
    return isinstance(m, QuantizedLinear)


def _is_gba_linear_or_switch(m) -> bool:
    from .switch_layers import QuantizedSwitchLinear          # late import: switch_layers imports ops like this module
    return isinstance(m, (QuantizedLinear, QuantizedSwitchLinear))


def _mlx_named_modules(model: nn.Module):
    """``(name, module)`` in the order MLX's ``Module.named_modules()`` yields them [external: mlx.nn.Module.apply_to_modules
    is a stack -- pop the last pushed entry, push the children in attribute order], i.e. depth first with siblings visited
    LAST to first.  Only the order matters here (the strategy loop's carry-over); names are torch's."""
    stack = [("", model)]
    while stack:
        prefix, mod = stack.pop()
        yield prefix, mod
        for cname, child in mod.named_children():
            stack.append((f"{prefix}.{cname}" if prefix else cname, child))


class QuantizedLinear(nn.Module):
    """y = x @ dequant(qweight, scales, zeros)^T (+ bias), weights packed at 2/3/4 bits."""

    def __init__(self, input_dims: int, output_dims: int, bias: bool = False, group_size: int = 64, bits: int = 4):
        super().__init__()
        self.group_size = group_size
        self.bits = bits
        self.output_dims = output_dims
        self.input_dims = input_dims
        self.double_group_size = 32
        if bias:
            self.register_buffer("bias", torch.zeros((output_dims,), dtype=torch.float16))
        else:
            self.bias = None
        self.init_params(False, False)

    # ------------------------------------------------------------------ parameters (:60-117)
    def _set(self, name: str, value: Optional[torch.Tensor]) -> None:
        if name in self._buffers:
            self._buffers[name] = value
        else:
            if hasattr(self, name):
                delattr(self, name)
            self.register_buffer(name, value)

    def init_params(self, use_double_quantization: bool, use_q_perm: bool) -> None:
        dev = self.qweight.device if getattr(self, "qweight", None) is not None else None
        shape_w = (self.output_dims, self.input_dims // 32 * self.bits)
        shape_sz = (self.output_dims, self.input_dims // self.group_size)
        self._set("qweight", torch.zeros(shape_w, dtype=torch.uint32, device=dev))
        # allocated by the reference but unused since 0.4.1 (:179-192): accepted on load, never applied
        self._set("channel_scale", torch.ones((1, 1, self.input_dims), dtype=torch.float16, device=dev))
        if use_q_perm:
            self._set("q_perm", torch.zeros((1, 1, self.input_dims), dtype=torch.int16, device=dev))
        if use_double_quantization:
            kg = math.ceil(self.input_dims / self.group_size)
            ng = math.ceil(self.output_dims / self.double_group_size)
            self._set("qstatistic", torch.zeros((kg, ng, self.double_group_size), dtype=torch.uint8, device=dev))
            self._set("qzeros_zeros", torch.zeros((kg, ng, 1), dtype=torch.float16, device=dev))
            self._set("qzeros_scales", torch.ones((kg, ng, 1), dtype=torch.float16, device=dev))
            self._set("qscales_zeros", torch.zeros((kg, ng, 1), dtype=torch.float16, device=dev))
            self._set("qscales_scales", torch.ones((kg, ng, 1), dtype=torch.float16, device=dev))
            self._set("scales", None)
            self._set("zeros", None)
        else:
            self._set("scales", torch.ones(shape_sz, dtype=torch.float16, device=dev))
            self._set("zeros", torch.zeros(shape_sz, dtype=torch.float16, device=dev))

    def create_scales_zeros(self) -> None:
        """GBA double-quant expansion -> fp16 scales/zeros [N, K/g] (:119-149), torch-CPU fp16 like the reference."""
        dev = self.qstatistic.device
        qstat = self.qstatistic.cpu()
        f16 = lambda t: t.cpu().to(torch.float16)
        buffer_shape = (math.ceil(self.input_dims / self.group_size), self.output_dims)
        qscales = (qstat & 0xF0) >> 4
        qzeros = qstat & 0x0F
        zeros = ((qzeros.to(torch.float16) - f16(self.qzeros_zeros)) * f16(self.qzeros_scales)).view(buffer_shape)
        scales = ((qscales.to(torch.float16) - f16(self.qscales_zeros)) * f16(self.qscales_scales)).view(buffer_shape)
        self._set("scales", scales.t().contiguous().to(dev))
        self._set("zeros", zeros.t().contiguous().to(dev))
        for nm in _DQ_NAMES:
            self._set(nm, None)

    def set_bias_and_weight(self) -> None:
        """GBA-native ``w = q*s - z``  ->  runtime ``w = q*s + zeros`` (:151-160)."""
        self._set("zeros", -self.zeros)
        if getattr(self, "q_perm", None) is not None:
            self._set("q_perm", self.q_perm.reshape(1, 1, -1))

    def extra_repr(self) -> str:
        return (f"input_dims={self.input_dims}, output_dims={self.output_dims}, bias={self.bias is not None}, "
                f"group_size={self.group_size}, bits={self.bits}")

    # ------------------------------------------------------------------ forward (:179-207)
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.scales is None or self.zeros is None:
            raise RuntimeError("QuantizedLinear: double-quantised statistics not expanded; call prepare_scales_zeros")
        # channel_scale / q_perm are intentionally NOT applied (commented out in the reference, :187-192)
        return ops.quantized_matmul(x, self.qweight, scales=self.scales, biases=self.zeros, transpose=True,
                                    group_size=self.group_size, bits=self.bits, bias=self.bias)

    # ------------------------------------------------------------------ loader hooks (:210-324)
    @classmethod
    def reinit_module(cls, model: nn.Module, group_size: int = 64, bits: int = 4, strategy: dict = None,
                      use_double_quantization: bool = False, use_q_perm: bool = False,
                      gba_linear_class_predicate: Callable = _is_gba_linear) -> None:
        """Per-layer mixed-bpw dispatch source: set (bits, group_size) on every QuantizedLinear and re-allocate
        its packed tensors.  With ``strategy`` (= quant_strategy.json["measurement"], utils.py:789-790) the entry
        is ``strategy["model.layers.<i>"][key]`` with ``key`` found by substring scan in the reference's fixed
        order, falling through on KeyError (:256-276)."""
        if strategy is None:
            for m in model.modules():
                if gba_linear_class_predicate(m):
                    m.group_size = group_size
                    m.bits = bits
                    m.init_params(use_double_quantization, use_q_perm)
            return
        from .switch_layers import QuantizedSwitchLinear
        last_strg = None
        for name, child in _mlx_named_modules(model):
            if isinstance(child, QuantizedSwitchLinear):
                # stacked experts: strategy["model.layers.<i>"]["moe_expert_<proj>"] over gate/up/down only (:236-252)
                per_block = strategy["model.layers.{}".format(name.split('.')[2])]
                strg = None
                for key in ('gate_proj', 'up_proj', 'down_proj'):
                    if key in name:
                        try:
                            strg = per_block['moe_expert_' + key]
                            break
                        except KeyError:
                            pass
                if strg is None:
                    raise KeyError(f"quant strategy has no moe_expert entry for {name}")
                child.bits = strg["bits"][0]
                child.group_size = strg["group_size"][str(child.bits)]
                assert child.group_size in [32, 64, 128], \
                    f"The group size value ({child.group_size}) must be 32, 64 or 128."
                child.init_params()
                last_strg = strg
                continue
            if not isinstance(child, QuantizedLinear):
                continue
            layer_number = name.split('.')[2]
            strategy_per_block = strategy["model.layers.{}".format(layer_number)]
            strg = None
            for key in _STRATEGY_KEYS:
                if key in name:
                    try:
                        if "shared_expert" in name:
                            key = 'moe_shared_expert_' + key
                        strg = strategy_per_block[key]
                        break
                    except KeyError:
                        pass
            if strg is None:
                # A name none of the keys matches -- the MoE router ``mlp.gate`` -- keeps the entry of the module visited
                # just before it in the reference (its loop variable simply carries over, :256-272).  "Before it" is in
                # MLX's traversal order (_mlx_named_modules): siblings last-to-first, so the router follows
                # ``switch_mlp.gate_proj`` and inherits ``moe_expert_gate_proj``.  Checkpoints are laid out accordingly, so
                # the carry-over is part of the format (load_model additionally reconciles it with the stored shapes).
                strg = last_strg
                child._strategy_carried_over = True
            if strg is None:
                raise KeyError(f"quant strategy has no entry for {name}")
            last_strg = strg
            child.bits = strg["bits"][0]
            child.group_size = strg["group_size"][str(child.bits)]
            assert child.group_size in [32, 64, 128], \
                f"The group size value ({child.group_size}) must be 32, 64 or 128."
            child.init_params(use_double_quantization, use_q_perm)

    @classmethod
    def prepare_scales_zeros(cls, model: nn.Module, gba_linear_class_predicate: Callable = _is_gba_linear) -> None:
        """Creates fp16 scales and zeros, releases double quantization parameters (:286-300)."""
        for m in model.modules():
            if gba_linear_class_predicate(m):
                m.create_scales_zeros()

    @classmethod
    def post_processing_and_release(cls, model: nn.Module,
                                    gba_linear_class_predicate: Callable = _is_gba_linear_or_switch) -> None:
        """zeros -> -zeros for GBA-native checkpoints (:303-324)."""
        for m in model.modules():
            if gba_linear_class_predicate(m):
                m.set_bias_and_weight()

    @classmethod
    def from_linear(cls, linear_layer: nn.Module, group_size: int = 64, bits: int = 4, q_perm=None,
                    channel_scale=None):
        raise NotImplementedError(
            "QuantizedLinear.from_linear needs mx.quantize (quantisation is outside the QuantizedLinear hot path)")
