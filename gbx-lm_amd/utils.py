"""``load`` / ``generate`` / ``stream_generate`` / ``generate_step`` for the MI355X QuantizedLinear path.

Host-side counterpart of gbx_lm/utils.py: same names, positional order and defaults
(load :882-888, generate :629-637, stream_generate :525-532, generate_step :217-232), PyTorch-ROCm
instead of MLX.  Options that would silently change behaviour (KV-cache quantisation, rotating cache,
speculative decoding, LoRA adapters) raise NotImplementedError.
"""
from __future__ import annotations

import glob
import json
import logging
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Callable, Generator, List, Optional, Tuple, Type, Union

import torch
from torch import nn

from .models import qdeepseek_v3, qllama, qmixtral, qphi3, qqwen2, qqwen3, qqwen3_moe
from .models import cache as cache_mod
from .models.quantized_linear_gba import QuantizedLinear
from .tokenizer_utils import TokenizerWrapper, load_tokenizer

MODEL_MAPPING = {
    "llama": qllama,
    "mistral": qllama,  # utils.py:52-61 maps mistral onto the llama graph
    "deepseek_v3": qdeepseek_v3,
    "mixtral": qmixtral,
    "phi3": qphi3,
    "qwen2": qqwen2,
    "qwen3": qqwen3,
    "qwen3_moe": qqwen3_moe,
}

_DQ_KEYS = ('qstatistic', 'qscales_scales', 'qzeros_scales', 'qscales_zeros', 'qzeros_zeros')


class ModelNotFoundError(Exception):
    def __init__(self, message):
        self.message = message
        super().__init__(self.message)


@dataclass
class GenerationResponse:
    """Output of :func:`stream_generate` (utils.py:75-104)."""
    text: str
    token: int
    logprobs: torch.Tensor
    from_draft: bool
    prompt_tokens: int
    prompt_tps: float
    generation_tokens: int
    generation_tps: float
    peak_memory: float
    finish_reason: Optional[str] = None
    hidden_states: Optional[torch.Tensor] = None


# --------------------------------------------------------------------------------------------- load
def _get_classes(config: dict):
    model_type = config["model_type"]
    if model_type not in MODEL_MAPPING:
        msg = f"Model type {model_type} not supported."
        logging.error(msg)
        raise ValueError(msg)
    arch = MODEL_MAPPING[model_type]
    return arch.Model, arch.ModelArgs


def get_model_path(path_or_hf_repo: str, token=None, revision: Optional[str] = None) -> Path:
    """Local directory, else HF snapshot (utils.py:162-201); offline boxes only have the first."""
    model_path = Path(path_or_hf_repo)
    if not model_path.exists():
        try:
            from huggingface_hub import snapshot_download
            model_path = Path(snapshot_download(
                repo_id=path_or_hf_repo, revision=revision, token=token,
                allow_patterns=["*.json", "model*.safetensors", "*.py", "tokenizer.model", "*.tiktoken", "*.txt"]))
        except Exception:
            raise ModelNotFoundError(
                f"Model not found for path or HF repo: {path_or_hf_repo}.\n"
                "Please make sure you specified the local path or Hugging Face repo id correctly.") from None
    return model_path


def get_parameter_usage_info(weights) -> Tuple[bool, bool]:
    """(use_double_quantization, use_q_perm) from the checkpoint keys (utils.py:696-719)."""
    use_double_quantization = False
    use_q_perm = False
    for k in weights:
        if any(s in k for s in _DQ_KEYS):
            use_double_quantization = True
        if 'q_perm' in k:
            use_q_perm = True
    return use_double_quantization, use_q_perm


def default_device() -> torch.device:
    if torch.cuda.is_available():
        import os
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device("cpu")


def load_weights(model: nn.Module, weights: dict, device=None) -> List[str]:
    """``model.load_weights(list(weights.items()), strict=False)`` (utils.py:862): keys with no matching
    tensor in the module tree are ignored (returned for inspection); shapes of matching tensors must agree."""
    modules = dict(model.named_modules())
    skipped = []
    for key, value in weights.items():
        mod_name, _, attr = key.rpartition('.')
        mod = modules.get(mod_name)
        if mod is None or not (attr in mod._buffers or attr in mod._parameters):
            skipped.append(key)
            continue
        cur = mod._buffers.get(attr) if attr in mod._buffers else mod._parameters.get(attr)
        if cur is not None and tuple(cur.shape) != tuple(value.shape):
            raise ValueError(f"shape mismatch for {key}: checkpoint {tuple(value.shape)} vs module {tuple(cur.shape)}")
        if attr == "qweight" and value.dtype == torch.int32:
            value = value.view(torch.uint32)
        value = value.to(device) if device is not None else value
        if attr in mod._buffers:
            mod._buffers[attr] = value
        else:
            mod._parameters[attr] = nn.Parameter(value, requires_grad=False)
    return skipped


def _reconcile_with_checkpoint(model, weights, use_double_quantization: bool, use_q_perm: bool) -> None:
    """A module whose strategy entry was only CARRIED OVER by the reference's loop (a name no key matches: the MoE router)
    gets whatever the traversal order happened to leave behind; the stored tensors are the ground truth there.  Where such a
    module disagrees with the shapes in the checkpoint it is re-allocated from them -- ``qweight [N, K*bits/32]``,
    ``scales [N, K/g]`` (``qstatistic [K/g, ...]`` when double-quantised).  Modules with a strategy entry of their own keep
    the reference's behaviour: a mismatch is an error."""
    for name, m in model.named_modules():
        if not isinstance(m, QuantizedLinear) or not getattr(m, "_strategy_carried_over", False):
            continue
        qw = weights.get(name + ".qweight")
        if qw is None or qw.dim() != 2 or qw.shape[0] != m.output_dims:
            continue
        k = m.input_dims
        bits = qw.shape[1] * 32 // k if k else 0
        gsrc = weights.get(name + ".qstatistic") if use_double_quantization else weights.get(name + ".scales")
        groups = (gsrc.shape[0] if use_double_quantization else gsrc.shape[-1]) if gsrc is not None else 0
        g = k // groups if groups else m.group_size
        if bits in (2, 3, 4) and qw.shape[1] * 32 == k * bits and g in (32, 64, 128) and (bits, g) != (m.bits, m.group_size):
            logging.warning(f"{name}: strategy says ({m.bits} bits, g{m.group_size}), checkpoint holds ({bits} bits, g{g}); using the checkpoint")
            m.bits, m.group_size = bits, g
            m.init_params(use_double_quantization, use_q_perm)


def load_model(model_path: Path, lazy: bool = False, model_config: dict = {}, bits: int = 4, group_size: int = 64,
               is_conversion: bool = False,
               get_model_classes: Callable[[dict], Tuple[Type[nn.Module], Type]] = _get_classes,
               device=None):
    """Build the model from ``config.json`` + ``quant_strategy.json`` + ``model*.safetensors`` (utils.py:752-879).

    Returns ``(model, config)``.  ``lazy`` is accepted for signature parity; tensors are always materialised
    on ``device`` (default: this rank's GPU if one is visible).
    """
    from safetensors.torch import load_file

    model_path = Path(model_path)
    device = torch.device(device) if device is not None else default_device()

    strategy = None
    try:
        with open(model_path / "quant_strategy.json", "r") as f:
            strategy = json.load(f)["measurement"]
    except FileNotFoundError:
        logging.info(f"[WARNING] Strategy config file not found in {model_path}")

    try:
        with open(model_path / "config.json", "r") as f:
            config = json.load(f)
            quantization = config.get("quantization", None)
            if quantization is None:
                quantization = {"group_size": group_size, "bits": bits}
    except FileNotFoundError:
        logging.info(f"[WARNING] Quantization config file not found in {model_path}")
        raise
    config.update(model_config)

    weight_files = glob.glob(str(model_path / "model*.safetensors"))
    if not weight_files:
        logging.error(f"No safetensors found in {model_path}")
        raise FileNotFoundError(f"No safetensors found in {model_path}")
    weights = {}
    for wf in weight_files:
        weights.update(load_file(wf))

    use_double_quantization, use_q_perm = get_parameter_usage_info(weights)
    if not use_q_perm:
        assert quantization['group_size'] in [32, 64, 128], \
            f"The group size value ({quantization['group_size']}) must be 32, 64 or 128."

    if is_conversion:   # GBA-native layout -> runtime layout (utils.py:828-838)
        for k, v in list(weights.items()):
            if 'qweight' in k:
                weights[k] = v.t().contiguous().view(torch.uint32) if v.dtype == torch.int32 else v.t().contiguous()
            if not use_double_quantization and ('scales' in k or 'zeros' in k):
                weights[k] = v.t().contiguous().to(torch.bfloat16)
            if any(s in k for s in ("norm.weight", "bias", "gate.weight", "lm_head", "embed_tokens", "channel_scale")):
                weights[k] = weights[k].to(torch.bfloat16)

    for k, v in list(weights.items()):   # utils.py:841-843
        if ("scale" in k or "zeros" in k) and not any(s in k for s in _DQ_KEYS):
            weights[k] = v.to(torch.bfloat16)

    model_class, model_args_class = get_model_classes(config=config)
    model_args = model_args_class.from_dict(config)
    model = model_class(model_args)

    if hasattr(model, "sanitize"):
        weights = model.sanitize(weights)

    QuantizedLinear.reinit_module(model, **quantization, strategy=strategy,
                                  use_double_quantization=use_double_quantization, use_q_perm=use_q_perm)
    _reconcile_with_checkpoint(model, weights, use_double_quantization, use_q_perm)
    load_weights(model, weights, device=device)
    model.to(device)

    if use_double_quantization:
        QuantizedLinear.prepare_scales_zeros(model)
    if is_conversion:
        QuantizedLinear.post_processing_and_release(model)

    # one-time dtype alignment: scales/zeros/bias follow the activation (embedding) dtype so the per-call
    # path never casts
    act_dtype = model.model.embed_tokens.weight.dtype if hasattr(model, "model") else torch.bfloat16
    if act_dtype in (torch.float16, torch.bfloat16):
        from .models.switch_layers import QuantizedSwitchLinear
        for m in model.modules():
            if isinstance(m, (QuantizedLinear, QuantizedSwitchLinear)):
                for nm in ("scales", "zeros", "bias"):
                    t = getattr(m, nm, None)
                    if t is not None and t.dtype != act_dtype:
                        m._buffers[nm] = t.to(act_dtype)
    model.eval()
    return model, config


def load(path_or_hf_repo: str, tokenizer_config={}, model_config={}, adapter_path: Optional[str] = None,
         lazy: bool = False) -> Tuple[nn.Module, TokenizerWrapper]:
    """Model + tokenizer from a local path or HF repo (utils.py:882-920)."""
    model_path = get_model_path(path_or_hf_repo)
    model, config = load_model(model_path, lazy, model_config=model_config)
    if adapter_path is not None:
        raise NotImplementedError("LoRA adapters are outside the QuantizedLinear hot path")
    tokenizer = load_tokenizer(model_path, tokenizer_config, eos_token_ids=config.get("eos_token_id", None))
    return model, tokenizer


# --------------------------------------------------------------------------------------------- generate
def _model_device(model) -> torch.device:
    return next(model.parameters()).device


def generate_step(prompt: torch.Tensor, model: nn.Module, *, max_tokens: int = 256,
                  sampler: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                  logits_processors: Optional[List[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]]] = None,
                  max_kv_size: Optional[int] = None, prompt_cache: Optional[Any] = None,
                  prefill_step_size: int = 2048, kv_bits: Optional[int] = None, kv_group_size: int = 64,
                  quantized_kv_start: int = 0, prompt_progress_callback: Optional[Callable[[int, int], None]] = None,
                  with_hidden_states: bool = False) -> Generator[Tuple[int, torch.Tensor, Optional[torch.Tensor]], None, None]:
    """Token generator (utils.py:217-338): chunked prefill, then one model call per token with B = 1.

    Yields ``(token: int, logprobs[vocab], hidden_states | None)``; ``logprobs = logits - logsumexp(logits)``
    (:305), default sampler greedy argmax (:285).
    """
    if kv_bits is not None:
        raise NotImplementedError("KV-cache quantisation (kv_bits) is outside the QuantizedLinear hot path")
    y = torch.as_tensor(prompt, dtype=torch.long, device=_model_device(model)).reshape(-1)
    tokens = None

    if prompt_cache is None:
        prompt_cache = cache_mod.make_prompt_cache(model, max_kv_size=max_kv_size)
    elif len(prompt_cache) != len(model.layers):
        raise ValueError("Wrong number of layers in the prompt cache.")
    prompt_progress_callback = prompt_progress_callback or (lambda *_: None)
    default_sampler = sampler is None                     # greedy (utils.py:285): token and logprobs can come from one device op
    sampler = sampler or (lambda x: torch.argmax(x, dim=-1))
    fast_decode = getattr(model, "decode_step", None) if not logits_processors else None
    # default sampler, no processors: the whole step -- blocks, head, logprobs, argmax, the token fed back on the device -- is one graph
    # replay, and (token, hand-off status words) reach the host in one copy (decode.FusedDecodeRunner.step_gen)
    fast_logprobs = getattr(model, "decode_step_logprobs", None) if (fast_decode is not None and default_sampler) else None
    step_words = [None]                                   # packed (token, status words) of the last fast_logprobs step (a static device buffer)

    @torch.no_grad()
    def _step(y, hidden_states=False):
        nonlocal tokens
        prompt_hidden_states = None
        step_words[0] = None
        if fast_logprobs is not None and y.numel() == 1 and not hidden_states:
            out = fast_logprobs(y, prompt_cache)
            if out is not None:
                step_words[0] = out[2]
                return out[0], out[1], None
        if fast_decode is not None and y.numel() == 1 and not hidden_states:
            logits = fast_decode(y, prompt_cache)                       # [1, vocab]
        elif hidden_states:
            logits, prompt_hidden_states = model(y[None], cache=prompt_cache, hidden_states=True)
            logits = logits[:, -1, :]
        elif hasattr(model, "last_logits"):
            logits = None
            if hasattr(model, "prompt_step") and not logits_processors:
                logits = model.prompt_step(y, prompt_cache)           # short fresh prompt: one replayed graph (None: not applicable)
            if logits is None:
                logits = model.last_logits(y[None], prompt_cache)      # output head on the last position only
        else:
            logits = model(y[None], cache=prompt_cache)[:, -1, :]
        if logits_processors:
            tokens = torch.cat([tokens, y]) if tokens is not None else y
            for processor in logits_processors:
                logits = processor(tokens, logits)
        if default_sampler and logits.is_cuda and logits.shape[0] == 1 and logits.dtype in (torch.bfloat16, torch.float16):
            from . import ops                                     # logsumexp, subtraction and argmax in two small launches
            logprobs, y = ops.logprobs_argmax(logits)
            return y.reshape(-1), logprobs.squeeze(0), prompt_hidden_states
        logprobs = logits - torch.logsumexp(logits.float(), dim=-1, keepdim=True).to(logits.dtype)
        y = sampler(logprobs)
        return y.reshape(-1), logprobs.squeeze(0), prompt_hidden_states

    with torch.no_grad():
        total_prompt_tokens = y.numel()
        prompt_processed_tokens = 0
        while y.numel() > prefill_step_size:
            if hasattr(model, "prefill"):
                model.prefill(y[:prefill_step_size][None], prompt_cache)    # these logits are never read: body only
            else:
                model(y[:prefill_step_size][None], cache=prompt_cache)
            prompt_progress_callback(prompt_processed_tokens, total_prompt_tokens)
            prompt_processed_tokens += prefill_step_size
            y = y[prefill_step_size:]
        first_in, first_off = y, (prompt_cache[0].offset if prompt_cache else 0)
        y, logprobs, h_states = _step(y, hidden_states=with_hidden_states)

    # One-token lookahead like the reference's mx.async_eval (utils.py:323-338): step n+1 is enqueued BEFORE token n is read on the
    # host.  On one in-order stream a plain ``y.item()`` would queue its copy behind step n+1 and wait for it, so the id goes to a
    # pinned host word right after its own step (async copy + event) and the host waits for that event only.
    # The status words of the decode runner's fused launches (in-launch hand-offs that can time out: another tenant on the GPU, a CU
    # mask) travel with the id: a non-zero word means the step's stream was poisoned (NaN logits) -- the model then falls back to the
    # five-launch block for good and the step is REPLAYED, so the caller never sees a token of a faulted step.
    ring, fring, pring, slot = [], [], [], [0]
    fault_words = getattr(model, "decode_fault_words", None) if fast_decode is not None else None

    class _HandoffFault(Exception):
        pass

    def _to_host(t):
        if not t.is_cuda:
            return t, None, None
        if not ring:
            ring.extend(torch.empty((1,), dtype=torch.long, pin_memory=True) for _ in range(3))   # (two ids are in flight at most)
            fring.extend(torch.zeros((4,), dtype=torch.int32, pin_memory=True) for _ in range(3))
        pinned, fpin = ring[slot[0] % 3], None
        if step_words[0] is not None:                     # one copy: (token, status words) as the step's last launch packed them
            if not pring:
                pring.extend(torch.zeros((4,), dtype=torch.long, pin_memory=True) for _ in range(3))
            pinned = pring[slot[0] % 3]
            pinned.copy_(step_words[0], non_blocking=True)
            fpin = pinned[1:]
        else:
            pinned.copy_(t.reshape(-1)[:1], non_blocking=True)
            words = fault_words() if fault_words is not None else []
            if words:
                fpin = fring[slot[0] % 3][:len(words)]
                for i, w in enumerate(words):
                    fpin[i:i + 1].copy_(w, non_blocking=True)
        slot[0] += 1
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
        return pinned, ev, fpin

    def _read(pending):
        host, ev, fpin = pending
        if ev is not None:
            ev.synchronize()
        if fpin is not None and bool(fpin.any()):
            raise _HandoffFault(int(fpin[fpin != 0][0]))
        return int(host[0])

    n = 0
    # (input, cache offset before it) of the decode step that produced ``y``; None when ``y`` came out of the prompt pass
    made_by = (first_in, first_off) if (first_in is not None and first_in.numel() == 1 and fast_decode is not None) else None
    pending = _to_host(y)
    while True:
        if n != max_tokens:
            ahead_off = prompt_cache[0].offset if prompt_cache else 0
            next_y, next_logprobs, _ = _step(y)
            next_pending = _to_host(next_y)
        if n == 0:
            prompt_progress_callback(total_prompt_tokens, total_prompt_tokens)
        if n == max_tokens:
            break
        try:
            tok = _read(pending)
        except _HandoffFault as fault:
            # in flight: the step that produced ``y`` (when it was a decode step) and the look-ahead step fed by it.  Either may be the
            # faulted one (the words are sticky): both are replayed on the five-launch block, from the cache offset they started at.
            model.decode_fallback(int(fault.args[0]), prompt_cache, made_by[1] if made_by is not None else ahead_off)
            if made_by is not None:
                src = made_by[0]                          # (an int: the token was read on the host; the device buffer it came from is reused)
                y, logprobs, h_states = _step(src if torch.is_tensor(src) else torch.tensor([src], dtype=torch.long, device=y.device))
            pending = _to_host(y)
            if n != max_tokens:
                ahead_off = prompt_cache[0].offset if prompt_cache else 0
                next_y, next_logprobs, _ = _step(y)
                next_pending = _to_host(next_y)
            try:
                tok = _read(pending)
            except _HandoffFault as again:
                raise RuntimeError(f"decode hand-off {again.args[0]} timed out again after the fall-back to the five-launch block") from None
        yield tok, logprobs, h_states
        made_by = (tok, ahead_off)                        # ``tok`` IS the value of ``y``, the input of the look-ahead step
        y, logprobs, pending = next_y, next_logprobs, next_pending
        n += 1


def stream_generate(model: nn.Module, tokenizer, prompt: Union[str, torch.Tensor, List[int]],
                    draft_model: Optional[nn.Module] = None, with_hidden_states: bool = False,
                    **kwargs) -> Generator[GenerationResponse, None, None]:
    """Text generator with throughput stats (utils.py:525-626)."""
    if draft_model is not None:
        raise NotImplementedError("speculative decoding is outside the QuantizedLinear hot path")
    if not isinstance(tokenizer, TokenizerWrapper):
        tokenizer = TokenizerWrapper(tokenizer)
    if not isinstance(prompt, torch.Tensor):
        if isinstance(prompt, str):
            add_special_tokens = tokenizer.bos_token is None or not prompt.startswith(tokenizer.bos_token)
            prompt = tokenizer.encode(prompt, add_special_tokens=add_special_tokens)
        prompt = torch.tensor(prompt, dtype=torch.long)
    detokenizer = tokenizer.detokenizer
    kwargs.pop("num_draft_tokens", None)
    token_generator = generate_step(prompt, model, with_hidden_states=with_hidden_states, **kwargs)

    dev = _model_device(model)
    def _sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    def _peak():
        return torch.cuda.max_memory_allocated(dev) / 1e9 if dev.type == "cuda" else 0.0

    detokenizer.reset()
    hidden_states = None
    tic = time.perf_counter()
    n, token, logprobs, prompt_tps = -1, None, None, 0.0
    for n, (token, logprobs, h_states) in enumerate(token_generator):
        if n == 0:
            hidden_states = h_states
            _sync()
            prompt_time = time.perf_counter() - tic
            prompt_tps = prompt.numel() / prompt_time
            tic = time.perf_counter()
        if token in tokenizer.eos_token_ids:
            break
        detokenizer.add_token(token)
        yield GenerationResponse(text=detokenizer.last_segment, token=token, logprobs=logprobs, from_draft=False,
                                 prompt_tokens=prompt.numel(), prompt_tps=prompt_tps, generation_tokens=n + 1,
                                 generation_tps=(n + 1) / (time.perf_counter() - tic), peak_memory=_peak(),
                                 finish_reason=None, hidden_states=hidden_states)
    if n < 0:
        return
    detokenizer.finalize()
    _sync()
    yield GenerationResponse(text=detokenizer.last_segment, token=token, logprobs=logprobs, from_draft=False,
                             prompt_tokens=prompt.numel(), prompt_tps=prompt_tps, generation_tokens=n + 1,
                             generation_tps=(n + 1) / (time.perf_counter() - tic), peak_memory=_peak(),
                             finish_reason="stop" if token in tokenizer.eos_token_ids else "length",
                             hidden_states=hidden_states)


def generate(model: nn.Module, tokenizer, prompt: Union[str, List[int]], verbose: bool = False,
             formatter: Optional[Callable] = None, with_hidden_states: bool = False, **kwargs):
    """Complete response as a string (utils.py:629-693)."""
    if formatter is not None:
        print("[Warning] Text formatting is deprecated and no longer used. "
              "The argument will be removed in a future version.")
    if verbose:
        print("=" * 10)
    all_hidden_states = [] if with_hidden_states else None
    hidden_states = None
    text = ""
    response = None
    for response in stream_generate(model, tokenizer, prompt, with_hidden_states=with_hidden_states, **kwargs):
        hidden_states = response.hidden_states
        if verbose:
            print(response.text, end="", flush=True)
        text += response.text
    if with_hidden_states and hidden_states is not None:
        all_hidden_states.append(hidden_states)
    if verbose:
        print()
        print("=" * 10)
        if len(text) == 0 or response is None:
            print("No text generated for this prompt")
            return
        print(f"Prompt: {response.prompt_tokens} tokens, {response.prompt_tps:.3f} tokens-per-sec")
        print(f"Generation: {response.generation_tokens} tokens, {response.generation_tps:.3f} tokens-per-sec")
        print(f"Peak memory: {response.peak_memory:.3f} GB")
    if with_hidden_states:
        return text, all_hidden_states
    return text
