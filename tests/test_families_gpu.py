"""GPU: Qwen2 / Qwen3 / Qwen3-MoE / Mixtral / Phi-3 / DeepSeek-V3 graphs on the HIP kernels against the SAME weights run through the CPU oracle
(prefill, then teacher-forced single-token steps; Qwen2's steps go through the graph-replayed generic decode runner)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("family", ["qwen2", "qwen3", "qwen3_moe", "mixtral", "phi3", "deepseek_v3"])
def test_family_gpu_matches_cpu_oracle(family, monkeypatch):
    import conftest
    from gbx_lm_amd import ops, synthetic as S
    from gbx_lm_amd.models.cache import make_prompt_cache
    cfg = {"qwen2": S.TINY_QWEN2, "qwen3": S.TINY_QWEN3, "qwen3_moe": S.TINY_MOE, "mixtral": S.TINY_MIXTRAL, "phi3": S.TINY_PHI3, "deepseek_v3": S.TINY_DEEPSEEK}[family]
    strat = S.make_strategy("S22", 2, 64)
    strat["model.layers.0"]["k_proj"] = {"bits": [2], "group_size": {"2": 32}}
    strat["model.layers.1"]["up_proj"] = {"bits": [3], "group_size": {"3": 128}}
    prompt = [11, 402, 7, 7, 93, 250, 3]
    forced = [5, 77, 140, 9]

    def run(model, dev):
        cache = make_prompt_cache(model)
        outs = []
        with torch.no_grad():
            lg = model(torch.tensor([prompt], device=dev), cache=cache)[:, -1, :]
            outs.append(lg.float().cpu())
            for t in forced:
                y = torch.tensor([t], device=dev)
                step = getattr(model, "decode_step", None)
                lg = step(y, cache) if step is not None else model(y[None], cache=cache)[:, -1, :]
                outs.append(lg.float().cpu().clone())
        return outs

    gpu_model = S.build_model(cfg, strat, device="cpu", seed=21).to(DEV)
    for blk in gpu_model.model.layers:
        blk.self_attn.rope.to(DEV)
    got = run(gpu_model, DEV)
    # single-token steps ran as HIP-graph replays: Qwen2 through the Llama-shaped runner (biased projections), the others
    # through the family-agnostic GraphDecodeRunner (static KV cache mode)
    assert getattr(gpu_model, "_decode_runner", None) is not None and gpu_model._decode_runner.graph is not None
    with monkeypatch.context() as mp:
        mp.setattr(ops, "quantized_matmul", conftest.oracle_matmul)
        mp.setattr(ops, "gather_quantized_matmul", conftest.oracle_gather_matmul)
        cpu_model = S.build_model(cfg, strat, device="cpu", seed=21)
        want = run(cpu_model, "cpu")
    for i, (g, w) in enumerate(zip(got, want)):
        err = (g - w).abs().max().item()
        assert err <= 0.03 * w.abs().max().item() + 0.02, (family, i, err, w.abs().max().item())
