"""GBA-native checkpoint -> runtime ("-mlx") layout, the role of gbx_lm/gba2mlx.py:93-161.

A GreenBitAI HF repo stores the packed tensors the way the GBA training stack does: ``qweight [K*bits/32, N]`` (the
transpose of what the runtime op streams), ``scales`` / ``zeros`` ``[K/g, N]`` in fp16 -- or, double-quantised, a uint8
``qstatistic`` + four fp16 side tensors -- and the zero with the GBA sign (``w = q*s - z``).  ``convert`` loads such a repo
through ``load_model(is_conversion=True)`` (transpose, double-quant expansion, sign flip: gbx_lm/utils.py:828-838,864-873,
gbx_lm/models/quantized_linear_gba.py:119-160), then writes the runtime layout back as safetensors shards + index +
config.json (with its ``quantization`` entry) + quant_strategy.json + tokenizer, after which plain ``load(mlx_path)`` works.
No network here: ``hf_path`` is a local directory (or an already cached snapshot); ``upload_repo`` raises.
"""
from __future__ import annotations

import argparse
import copy
import json
import re
import shutil
from pathlib import Path
from typing import Dict, Optional, Tuple, Union

import torch

MAX_FILE_SIZE_GB = 5


def configure_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Convert a GBA-native Hugging Face model to the runtime (MLX) layout")
    parser.add_argument("--hf-path", type=str, help="Path to the Hugging Face model.")
    parser.add_argument("--mlx-path", type=str, default="mlx_model", help="Path to save the converted model.")
    parser.add_argument("--q-group-size", help="Group size used in the quantized model", type=int, default=256)
    parser.add_argument("--q-bits", help="Bits per weight for quantization.", type=int, default=4)
    parser.add_argument("--hf-token", type=str, default="", help="hf token")
    parser.add_argument("--upload-repo", help="The Hugging Face repo to upload the model to.", type=str, default=None)
    return parser


def extract_bits_and_group_size(s: str) -> Tuple[Optional[int], Optional[int]]:
    """``w<bits>`` / ``g<group>`` patterns of a GreenBitAI repo name (gbx_lm/utils.py:722-749); None where absent."""
    b, g = re.search(r"w(\d+)", s), re.search(r"g(\d+)", s)
    return (int(b.group(1)) if b else None), (int(g.group(1)) if g else None)


def make_shards(weights: Dict[str, torch.Tensor], max_file_size_gb: int = MAX_FILE_SIZE_GB) -> list:
    """gbx_lm/utils.py:967-988: greedy split of the flat weight dict into files of at most ``max_file_size_gb``."""
    limit = max_file_size_gb << 30
    shards, shard, size = [], {}, 0
    for k, v in weights.items():
        nbytes = v.numel() * v.element_size()
        if size + nbytes > limit:
            shards.append(shard)
            shard, size = {}, 0
        shard[k] = v
        size += nbytes
    shards.append(shard)
    return shards


def save_weights(save_path: Union[str, Path], weights: Dict[str, torch.Tensor], *, donate_weights: bool = False) -> None:
    """gbx_lm/utils.py:1055-1104: ``model.safetensors`` or ``model-0000i-of-0000n.safetensors`` + ``model.safetensors.index.json``."""
    from safetensors.torch import save_file
    save_path = Path(save_path)
    save_path.mkdir(parents=True, exist_ok=True)
    shards = make_shards(weights)
    n = len(shards)
    fmt = "model-{:05d}-of-{:05d}.safetensors" if n > 1 else "model.safetensors"
    index = {"metadata": {"total_size": sum(v.numel() * v.element_size() for v in weights.values())}, "weight_map": {}}
    if donate_weights:
        weights.clear()
    for i in range(n):
        shard, shards[i] = shards[i], None
        name = fmt.format(i + 1, n)
        save_file({k: v.contiguous() for k, v in shard.items()}, str(save_path / name), metadata={"format": "mlx"})
        for k in shard:
            index["weight_map"][k] = name
    index["weight_map"] = {k: index["weight_map"][k] for k in sorted(index["weight_map"])}
    with open(save_path / "model.safetensors.index.json", "w") as f:
        json.dump(index, f, indent=4)


def save_config(config: dict, config_path: Union[str, Path]) -> None:
    """gbx_lm/utils.py:1107-1127."""
    config.pop("_name_or_path", None)
    with open(config_path, "w") as f:
        json.dump(dict(sorted(config.items())), f, indent=4)


def fetch_from_hub(model_path: str, lazy: bool = False, token=None, group_size: int = 64, bits: int = 4,
                   is_conversion: bool = False, device="cpu"):
    """gbx_lm/utils.py:923-964: ``(model, config, tokenizer, local path)``; the conversion itself happens in ``load_model``."""
    from .tokenizer_utils import load_tokenizer
    from .utils import get_model_path, load_model
    path = get_model_path(model_path, token=token)
    model, config = load_model(path, lazy, bits=bits, group_size=group_size, is_conversion=is_conversion, device=device)
    tokenizer = load_tokenizer(path, {}, eos_token_ids=config.get("eos_token_id", None))
    return model, config, tokenizer, path


def get_quantized_parameters(model, config: dict, q_group_size: int, q_bits: int):
    """gba2mlx.py:47-66: the flat runtime-layout tensors + the config with its ``quantization`` entry."""
    from .synthetic import state_dict_for_disk
    cfg = copy.deepcopy(config)
    if q_group_size is not None and q_bits is not None:
        cfg["quantization"] = {"group_size": q_group_size, "bits": q_bits}
    return state_dict_for_disk(model), cfg


def check_upload_repo_string(repo_string: str):
    example = "ExampleHFRepoName/example-folder-name-mlx"
    if "/" not in repo_string:
        return False, f"Upload repo string must contain a '/' separating the HF repo name and the folder name. For example, '{example}'."
    if not repo_string.split("/", 1)[1].endswith("-mlx"):
        return False, f"The folder name must end with '-mlx'. For example, '{example}'."
    return True, ""


def convert(hf_path: str, mlx_path: str = "mlx_model", q_group_size: int = 256, q_bits: int = 4, upload_repo: str = None,
            hf_token: str = None) -> None:
    """gba2mlx.py:93-161.  As there, ``w<bits>`` / ``g<group>`` in the repo name override the arguments; a name without them
    keeps the arguments (the reference would pass None on)."""
    if upload_repo is not None:
        raise NotImplementedError("uploading to the Hugging Face hub is outside the QuantizedLinear hot path (and this box has no network)")
    print("[INFO] Loading ...")
    nb, ng = extract_bits_and_group_size(str(hf_path))
    q_bits, q_group_size = (nb if nb is not None else q_bits), (ng if ng is not None else q_group_size)
    model, config, tokenizer, model_path = fetch_from_hub(hf_path, token=hf_token, group_size=q_group_size, bits=q_bits,
                                                          is_conversion=True)
    if "quantization" in config:                       # the repo's own entry wins over name / arguments, as load_model used it
        q_group_size, q_bits = config["quantization"]["group_size"], config["quantization"]["bits"]
    weights, config = get_quantized_parameters(model, config, q_group_size, q_bits)
    print(f"[INFO] Creating dir '{mlx_path}' for saving ...")
    mlx_path = Path(mlx_path)
    mlx_path.mkdir(parents=True, exist_ok=True)
    print("[INFO] Saving quantized weights ...")
    del model
    save_weights(mlx_path, weights, donate_weights=True)
    for f in ("tokenizer.json", "tokenizer_config.json", "tokenizer.model", "special_tokens_map.json", "vocab.json", "merges.txt"):
        if (Path(model_path) / f).exists():
            shutil.copy(Path(model_path) / f, mlx_path / f)
    print("[INFO] Saving config file ...")
    save_config(config, config_path=mlx_path / "config.json")
    print("[INFO] Saving quant strategy config file ...")
    if (Path(model_path) / "quant_strategy.json").exists():
        shutil.copy(Path(model_path) / "quant_strategy.json", mlx_path / "quant_strategy.json")
        print(f"[INFO] Strategy config file copied to {mlx_path}")
    else:
        print(f"[INFO] Strategy config file not found in {model_path}")
    print(f"[INFO] Converted model files have been saved in {mlx_path.absolute()}/")
    for file in sorted(mlx_path.iterdir()):
        if file.is_file():
            print("[INFO] |----" + file.name)


if __name__ == "__main__":
    convert(**vars(configure_parser().parse_args()))
