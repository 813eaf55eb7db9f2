"""Synthetic GBA checkpoints with the real tensor names / shapes / dtypes (no network, no real weights).

Distributions follow SURVEY.md 8d: uniform random packed words, ``scales ~ U(0.5,1.5)*2/(2^b-1)*0.02``,
``zeros = -scales*(2^b-1)/2 + N(0, 0.1*scales)`` (MLX-signed, i.e. added), embeddings ``N(0, 0.02)``.
Used by bench.py, the tests and ``__graft_entry__.smoke()``; the layout written to disk is exactly what
``load()`` reads (gbx_lm/utils.py:752-879 contract).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

from .models import qdeepseek_v3, qllama, qmixtral, qphi3, qqwen2, qqwen3, qqwen3_moe
from .models.switch_layers import QuantizedSwitchLinear
from .models.quantized_linear_gba import QuantizedLinear

PROJ = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")

LLAMA3_8B = dict(model_type="llama", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, vocab_size=128256,
                 rope_theta=500000.0, max_position_embeddings=8192, tie_word_embeddings=False,
                 eos_token_id=128009)
LLAMA32_3B = dict(model_type="llama", hidden_size=3072, intermediate_size=8192, num_hidden_layers=28,
                  num_attention_heads=24, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-5, vocab_size=128256,
                  rope_theta=500000.0, max_position_embeddings=131072, tie_word_embeddings=True,
                  rope_scaling=dict(factor=32.0, high_freq_factor=4.0, low_freq_factor=1.0,
                                    original_max_position_embeddings=8192, rope_type="llama3"),
                  eos_token_id=128009)
TINY = dict(model_type="llama", hidden_size=256, intermediate_size=512, num_hidden_layers=2,
            num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, vocab_size=512,
            rope_theta=10000.0, max_position_embeddings=512, tie_word_embeddings=False, eos_token_id=511)


QWEN3_30B_A3B = dict(model_type="qwen3_moe", hidden_size=2048, intermediate_size=6144, moe_intermediate_size=768,
                     num_hidden_layers=48, num_attention_heads=32, num_key_value_heads=4, head_dim=128, num_experts=128,
                     num_experts_per_tok=8, decoder_sparse_step=1, mlp_only_layers=[], norm_topk_prob=True,
                     rms_norm_eps=1e-6, vocab_size=151936, rope_theta=1000000.0, max_position_embeddings=40960,
                     tie_word_embeddings=False, eos_token_id=151645)          # public Qwen3-30B-A3B shapes
QWEN3_8B = dict(model_type="qwen3", hidden_size=4096, intermediate_size=12288, num_hidden_layers=36, num_attention_heads=32,
                num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6, vocab_size=151936, rope_theta=1000000.0,
                max_position_embeddings=40960, tie_word_embeddings=False, eos_token_id=151645)     # public Qwen3-8B shapes
TINY_MOE = dict(model_type="qwen3_moe", hidden_size=256, intermediate_size=512, moe_intermediate_size=128,
                num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=64, num_experts=8,
                num_experts_per_tok=2, decoder_sparse_step=1, mlp_only_layers=[], norm_topk_prob=True,
                rms_norm_eps=1e-6, vocab_size=512, rope_theta=10000.0, max_position_embeddings=512,
                tie_word_embeddings=False, eos_token_id=511)


TINY_QWEN2 = dict(model_type="qwen2", hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                  num_key_value_heads=2, rms_norm_eps=1e-6, vocab_size=512, rope_theta=1000000.0,
                  max_position_embeddings=512, tie_word_embeddings=True, eos_token_id=511)
TINY_MIXTRAL = dict(model_type="mixtral", hidden_size=256, intermediate_size=128, num_hidden_layers=2,
                    num_attention_heads=4, num_key_value_heads=2, num_local_experts=4, num_experts_per_tok=2,
                    rms_norm_eps=1e-5, vocab_size=512, rope_theta=1000000.0, max_position_embeddings=512, eos_token_id=511)
TINY_PHI3 = dict(model_type="phi3", hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                 num_key_value_heads=2, rms_norm_eps=1e-5, vocab_size=512, rope_theta=10000.0,
                 max_position_embeddings=4096, original_max_position_embeddings=512, eos_token_id=511,
                 rope_scaling={"type": "su", "short_factor": [1.0] * 32,
                               "long_factor": [1.0 + 0.05 * i for i in range(32)]})
TINY_DEEPSEEK = dict(model_type="deepseek_v3", vocab_size=512, hidden_size=256, intermediate_size=512, moe_intermediate_size=128,
                     num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, n_shared_experts=1, n_routed_experts=8,
                     routed_scaling_factor=2.5, kv_lora_rank=128, q_lora_rank=128, qk_rope_head_dim=32, v_head_dim=32,
                     qk_nope_head_dim=32, n_group=4, topk_group=2, num_experts_per_tok=2, first_k_dense_replace=1,
                     max_position_embeddings=512, rms_norm_eps=1e-6, rope_theta=10000.0, eos_token_id=511,
                     rope_scaling={"type": "yarn", "factor": 4.0, "original_max_position_embeddings": 128, "beta_fast": 32,
                                   "beta_slow": 1, "mscale": 1.0, "mscale_all_dim": 1.0})
TINY_QWEN3 = dict(model_type="qwen3", hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                  num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-6, vocab_size=512, rope_theta=1000000.0,
                  max_position_embeddings=512, tie_word_embeddings=False, eos_token_id=511)


def strategy_bits(name: str, layer: int, n_layers: int) -> int:
    """S40: all 4-bit.  S22: first/last layer 4-bit, layers 1,2 and n-2 3-bit, else 2-bit."""
    if name == "S40":
        return 4
    if name == "S22":
        if layer in (0, n_layers - 1):
            return 4
        if layer in (1, 2, n_layers - 2):
            return 3
        return 2
    if name == "S30":
        return 3
    if name == "S20":
        return 2
    raise ValueError(f"unknown synthetic strategy {name}")


def make_strategy(name: str, n_layers: int, group_size: int = 128) -> Dict:
    """quant_strategy.json["measurement"] in the reference's schema (quantized_linear_gba.py:256-272)."""
    strat = {}
    for i in range(n_layers):
        b = strategy_bits(name, i, n_layers)
        strat[f"model.layers.{i}"] = {p: {"bits": [b], "group_size": {str(b): group_size}} for p in PROJ}
        for p in ("kv_a_proj_with_mqa", "kv_b_proj", "q_a_proj", "q_b_proj",            # DeepSeek-V3 latent attention
                  "moe_shared_expert_gate_proj", "moe_shared_expert_up_proj", "moe_shared_expert_down_proj"):
            strat[f"model.layers.{i}"][p] = {"bits": [b], "group_size": {str(b): group_size}}
        for p in ("qkv_proj", "gate_up_proj"):      # Phi-3 fused projections; the reference's substring scan reaches them only
            strat[f"model.layers.{i}"][p] = {"bits": [b], "group_size": {str(b): group_size}}   # when v_proj / up_proj are absent
        # stacked experts (QuantizedSwitchLinear): 'moe_expert_<proj>', one step narrower than the dense projections
        be = max(2, b - 1)
        for p in ("gate_proj", "up_proj", "down_proj"):
            strat[f"model.layers.{i}"]["moe_expert_" + p] = {"bits": [be], "group_size": {str(be): group_size}}
    return strat


def _fill_qlinear(m: QuantizedLinear, gen: torch.Generator, device, act_dtype) -> None:
    n, k, b, g = m.output_dims, m.input_dims, m.bits, m.group_size
    words = k // 32 * b
    qw = torch.randint(0, 2**32, (n, words), dtype=torch.int64, device=device, generator=gen).to(torch.int32)
    qmax = float((1 << b) - 1)
    sc = (torch.rand((n, k // g), device=device, generator=gen) + 0.5) * (2.0 / qmax) * 0.02
    ze = -sc * qmax / 2.0 + torch.randn((n, k // g), device=device, generator=gen) * 0.1 * sc
    m._buffers["qweight"] = qw.view(torch.uint32)
    m._buffers["scales"] = sc.to(act_dtype)
    m._buffers["zeros"] = ze.to(act_dtype)
    m._buffers["channel_scale"] = torch.ones((1, 1, k), dtype=torch.float16, device=device)
    if m.bias is not None:
        m._buffers["bias"] = (torch.randn((n,), device=device, generator=gen) * 0.01).to(act_dtype)


def _fill_switch(m: QuantizedSwitchLinear, gen: torch.Generator, device, act_dtype) -> None:
    e, n, k, b, g = m.num_experts, m.output_dims, m.input_dims, m.bits, m.group_size
    qw = torch.randint(0, 2**32, (e, n, k // 32 * b), dtype=torch.int64, device=device, generator=gen).to(torch.int32)
    qmax = float((1 << b) - 1)
    sc = (torch.rand((e, n, k // g), device=device, generator=gen) + 0.5) * (2.0 / qmax) * 0.02
    ze = -sc * qmax / 2.0 + torch.randn((e, n, k // g), device=device, generator=gen) * 0.1 * sc
    m._buffers["qweight"] = qw.view(torch.uint32)
    m._buffers["scales"] = sc.to(act_dtype)
    m._buffers["zeros"] = ze.to(act_dtype)
    if m.bias is not None:
        m._buffers["bias"] = (torch.randn((e, n), device=device, generator=gen) * 0.01).to(act_dtype)


def build_model(config: Dict, strategy: Optional[Dict] = None, device="cpu", seed: int = 0,
                act_dtype=torch.bfloat16, bits: int = 4, group_size: int = 64) -> qllama.Model:
    """Random-init model with real shapes, generated directly on ``device`` (no files)."""
    device = torch.device(device)
    arch = {"qwen3_moe": qqwen3_moe, "qwen3": qqwen3, "qwen2": qqwen2, "mixtral": qmixtral, "phi3": qphi3, "deepseek_v3": qdeepseek_v3}.get(config.get("model_type"), qllama)
    args = arch.ModelArgs.from_dict(config)
    with torch.device("meta"):
        model = arch.Model(args)
    model = model.to_empty(device=device)
    QuantizedLinear.reinit_module(model, group_size=group_size, bits=bits, strategy=strategy)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    for m in model.modules():
        if isinstance(m, QuantizedLinear):
            for nm in list(m._buffers):
                if m._buffers[nm] is not None and m._buffers[nm].device != device:
                    m._buffers[nm] = torch.empty_like(m._buffers[nm], device=device)
            _fill_qlinear(m, gen, device, act_dtype)
        elif isinstance(m, QuantizedSwitchLinear):
            _fill_switch(m, gen, device, act_dtype)
        elif isinstance(m, qllama.RMSNorm):
            m.weight = torch.nn.Parameter(torch.ones(m.weight.shape, dtype=act_dtype, device=device), requires_grad=False)
        elif isinstance(m, torch.nn.Embedding):
            w = torch.randn(m.weight.shape, device=device, generator=gen) * 0.02
            m.weight = torch.nn.Parameter(w.to(act_dtype), requires_grad=False)
        elif isinstance(m, torch.nn.Linear):
            w = torch.randn(m.weight.shape, device=device, generator=gen) * 0.02
            m.weight = torch.nn.Parameter(w.to(act_dtype), requires_grad=False)
        elif isinstance(m, qdeepseek_v3.MoEGate):
            w = torch.randn(m.weight.shape, device=device, generator=gen) * 0.5
            m.weight = torch.nn.Parameter(w.to(act_dtype), requires_grad=False)
            m.e_score_correction_bias = torch.nn.Parameter(torch.randn(m.e_score_correction_bias.shape, device=device,
                                                                       generator=gen) * 0.1, requires_grad=False)
        elif hasattr(m, "inv_freq") and isinstance(getattr(m, "inv_freq"), torch.Tensor):
            pass
    # rope tables are non-persistent buffers created on meta: rebuild them for real
    from .models.rope_utils import initialize_rope
    for blk in model.model.layers:
        at = blk.self_attn
        if hasattr(at, "build_rope"):               # families with their own embedding (Phi-3 su / longrope)
            at.rope = at.build_rope(args).to(device)
            continue
        at.rope = initialize_rope(at.head_dim, args.rope_theta, getattr(args, "rope_traditional", False),
                                  getattr(args, "rope_scaling", None) if arch in (qllama, qqwen2) else None,
                                  args.max_position_embeddings).to(device)
    model.eval()
    return model


def state_dict_for_disk(model: qllama.Model) -> Dict[str, torch.Tensor]:
    """Checkpoint keys exactly as a ``-mlx`` GBA repo stores them."""
    out = {}
    for name, m in model.named_modules():
        if isinstance(m, QuantizedLinear):
            for nm in ("qweight", "scales", "zeros", "bias", "channel_scale"):
                t = getattr(m, nm, None)
                if t is not None:
                    out[f"{name}.{nm}"] = t.detach().cpu().contiguous()
        elif isinstance(m, QuantizedSwitchLinear):
            # on disk the experts are separate modules ``...mlp.experts.<e>.<proj>`` (Model.sanitize stacks them)
            base, proj = name.rsplit(".switch_mlp.", 1)
            if base.endswith("block_sparse_moe"):                   # Mixtral checkpoints call the projections w1 / w2 / w3
                proj = {"gate_proj": "w1", "down_proj": "w2", "up_proj": "w3"}[proj]
            for e in range(m.num_experts):
                for nm in ("qweight", "scales", "zeros"):
                    out[f"{base}.experts.{e}.{proj}.{nm}"] = getattr(m, nm)[e].detach().cpu().contiguous()
        elif isinstance(m, (qllama.RMSNorm, torch.nn.Embedding, torch.nn.Linear)):
            out[f"{name}.weight"] = m.weight.detach().cpu().contiguous()
        elif isinstance(m, qdeepseek_v3.MoEGate):
            out[f"{name}.weight"] = m.weight.detach().cpu().contiguous()
            out[f"{name}.e_score_correction_bias"] = m.e_score_correction_bias.detach().cpu().contiguous()
    return out


def write_tokenizer(path: str, vocab_size: int, eos_token_id: int) -> None:
    """Tiny offline WordLevel tokenizer so that ``load()`` / ``generate(str)`` run end to end."""
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders
    vocab = {f"w{i}": i for i in range(vocab_size)}
    vocab["<eos>"] = eos_token_id
    vocab = {k: v for k, v in vocab.items() if not (k == f"w{eos_token_id}")}
    vocab["hello"], vocab["world"] = vocab.pop("w1"), vocab.pop("w2")
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="w0"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.decoder = decoders.Fuse()
    tok.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "eos_token": "<eos>", "unk_token": "w0"}, f)


def write_checkpoint(path: str, config: Dict, strategy: Optional[Dict], seed: int = 0,
                     quantization: Optional[Dict] = None, extra_tensors: Optional[Dict] = None) -> qllama.Model:
    """config.json + quant_strategy.json + model.safetensors (+ tokenizer) for ``load(path)``."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    q = quantization or {"group_size": 64, "bits": 4}
    model = build_model(config, strategy, device="cpu", seed=seed, bits=q["bits"], group_size=q["group_size"])
    cfg = dict(config)
    cfg["quantization"] = q
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    if strategy is not None:
        with open(os.path.join(path, "quant_strategy.json"), "w") as f:
            json.dump({"measurement": strategy}, f)
    sd = state_dict_for_disk(model)
    if extra_tensors:
        sd.update(extra_tensors)
    save_file(sd, os.path.join(path, "model.safetensors"))
    write_tokenizer(path, config["vocab_size"], config.get("eos_token_id", config["vocab_size"] - 1))
    return model
