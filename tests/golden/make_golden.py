"""Generates the committed golden fixtures of tests/golden/ with OUR oracle (oracle/gba_oracle.py).

The reference cannot run here (it imports MLX on line 1 of every module: gbx_lm/__init__.py:1 ->
gbx_lm/utils.py:26) and holds no golden vectors for this path, so these are oracle outputs,
reviewed against the hand-checkable known-answer rows of SURVEY.md Appendix A (also written here as data).

    python tests/golden/make_golden.py          # rewrites the fixtures deterministically
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import gba_oracle as O  # noqa: E402


def known_answer():
    """(i) rows whose fields are k mod 2^b, k = 0..31 -- checkable by eye."""
    return {
        "2": {"words": ["0xE4E4E4E4", "0xE4E4E4E4"], "fields": [k % 4 for k in range(32)]},
        "3": {"words": ["0x88FAC688", "0xC688FAC6", "0xFAC688FA"], "fields": [k % 8 for k in range(32)],
              "straddlers": [10, 21]},
        "4": {"words": ["0x76543210", "0xFEDCBA98", "0x76543210", "0xFEDCBA98"], "fields": [k % 16 for k in range(32)]},
    }


def small_cases():
    """(ii) N=8, K=256 random cases per (bits, group), bf16 and f16, with bias on some."""
    out = {}
    for b in (2, 3, 4):
        for g in (32, 64, 128):
            for dt in ("bf16", "f16"):
                seed = 1000 * b + g + (7 if dt == "f16" else 0)
                bias = (g == 64)
                d = O.synth_layer(8, 256, b, g, seed=seed, act_dtype=dt, m=3, bias=bias)
                y64 = O.qlinear_ref(d["x"], d["qweight"], d["scales"], d["zeros"], b, g, bias=d["bias"], acc=np.float64)
                key = f"b{b}_g{g}_{dt}"
                view = (lambda t: t.view(torch.int16).numpy().view(np.uint16))
                out[key + "_qweight"] = d["qweight"]
                out[key + "_scales"] = view(d["scales"])
                out[key + "_zeros"] = view(d["zeros"])
                out[key + "_x"] = view(d["x"])
                if bias:
                    out[key + "_bias"] = view(d["bias"])
                out[key + "_q"] = O.unpack(d["qweight"], b).astype(np.int8)
                out[key + "_y64"] = y64
    return out


def double_quant_case():
    """(iii) [2,1,32] qstatistic fixture for create_scales_zeros (quantized_linear_gba.py:119-149)."""
    rng = np.random.default_rng(42)
    K, N, g = 256, 32, 128
    qstat = rng.integers(0, 256, size=(K // g, 1, 32), dtype=np.uint8)
    mk = lambda lo, hi: rng.uniform(lo, hi, size=(K // g, 1, 1)).astype(np.float16)
    qzz, qzs, qsz, qss = mk(6, 9), mk(0.01, 0.02), mk(-1, 1), mk(0.001, 0.003)
    scales, zeros = O.create_scales_zeros(qstat, qzz, qzs, qsz, qss, K, N, g)
    return dict(qstatistic=qstat, qzeros_zeros=qzz, qzeros_scales=qzs, qscales_zeros=qsz, qscales_scales=qss,
                scales=scales.numpy(), zeros=zeros.numpy(), K=K, N=N, g=g)


TINY_PROMPT = [5, 17, 42, 7, 99]
TINY_STEPS = 48


def tiny_strategy():
    from gbx_lm_amd import synthetic as S
    strat = S.make_strategy("S22", 2)                       # layer 0 and 1 -> 4-bit g128 (n_layers = 2)
    strat["model.layers.0"]["q_proj"] = {"bits": [3], "group_size": {"3": 32}}
    strat["model.layers.0"]["k_proj"] = {"bits": [2], "group_size": {"2": 64}}
    strat["model.layers.0"]["gate_proj"] = {"bits": [2], "group_size": {"2": 32}}
    strat["model.layers.1"]["down_proj"] = {"bits": [3], "group_size": {"3": 128}}
    strat["model.layers.1"]["o_proj"] = {"bits": [2], "group_size": {"2": 128}}
    return strat


def tiny_llama_case():
    """(iv) 2-layer hidden-256 synthetic Llama with a mixed 2/3/4-bit strategy: greedy ids + first-step logits,
    computed by the host graph with the QuantizedLinear op routed through the oracle on CPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import oracle_matmul
    from gbx_lm_amd import ops, synthetic as S, utils as U
    ops.quantized_matmul = oracle_matmul
    model = S.build_model(S.TINY, tiny_strategy(), device="cpu", seed=7)
    steps = list(U.generate_step(torch.tensor(TINY_PROMPT), model, max_tokens=TINY_STEPS))
    ids = [t for t, _, _ in steps]
    lps = torch.stack([lp.float() for _, lp, _ in steps]).numpy()
    top2 = np.sort(lps, axis=1)[:, -2:]
    return dict(prompt=np.array(TINY_PROMPT), ids=np.array(ids), logprobs=lps.astype(np.float32),
                gap=(top2[:, 1] - top2[:, 0]).astype(np.float32))


MOE_PROMPT = [3, 141, 59, 26, 5, 35]
MOE_STEPS = 24


def moe_strategy():
    from gbx_lm_amd import synthetic as S
    strat = S.make_strategy("S22", 2, 64)
    strat["model.layers.0"]["moe_expert_up_proj"] = {"bits": [2], "group_size": {"2": 32}}
    strat["model.layers.1"]["moe_expert_down_proj"] = {"bits": [4], "group_size": {"4": 128}}
    strat["model.layers.1"]["o_proj"] = {"bits": [3], "group_size": {"3": 64}}
    # (the routers match no strategy key and inherit moe_expert_gate_proj: MLX visits switch_mlp.gate_proj just before mlp.gate)
    return strat


def tiny_moe_case():
    """(vi) 2-layer Qwen3-MoE (8 experts, top-2, stacked QuantizedSwitchLinear at 2/3/4 bits): greedy ids + logprobs
    from the host graph with QuantizedLinear / gather_qmm routed through the oracle on CPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import oracle_matmul, oracle_gather_matmul
    from gbx_lm_amd import ops, synthetic as S, utils as U
    ops.quantized_matmul = oracle_matmul
    ops.gather_quantized_matmul = oracle_gather_matmul
    model = S.build_model(S.TINY_MOE, moe_strategy(), device="cpu", seed=11)
    steps = list(U.generate_step(torch.tensor(MOE_PROMPT), model, max_tokens=MOE_STEPS))
    ids = [t for t, _, _ in steps]
    lps = torch.stack([lp.float() for _, lp, _ in steps]).numpy()
    top2 = np.sort(lps, axis=1)[:, -2:]
    return dict(prompt=np.array(MOE_PROMPT), ids=np.array(ids), logprobs=lps.astype(np.float32),
                gap=(top2[:, 1] - top2[:, 0]).astype(np.float32))


def eminf_case():
    """(v) EMINF: a fixed [512] logits vector -> best_logits, best_entropy (oracle numpy restatement)."""
    rng = np.random.default_rng(123)
    logits = (rng.standard_normal(512) * 2.5).astype(np.float32)
    best, h = O.eminf_optimize(logits)
    sharp = logits.copy(); sharp[7] += 30.0                         # already below the entropy target: returns input
    b2, h2 = O.eminf_optimize(sharp)
    return dict(logits=logits, best=best, entropy=np.float32(h), sharp=sharp, sharp_best=b2, sharp_entropy=np.float32(h2))


def main():
    with open(os.path.join(HERE, "known_answer.json"), "w") as f:
        json.dump(known_answer(), f, indent=1)
    np.savez_compressed(os.path.join(HERE, "qlinear_small.npz"), **small_cases())
    np.savez_compressed(os.path.join(HERE, "double_quant.npz"), **double_quant_case())
    np.savez_compressed(os.path.join(HERE, "tiny_llama.npz"), **tiny_llama_case())
    np.savez_compressed(os.path.join(HERE, "eminf.npz"), **eminf_case())
    np.savez_compressed(os.path.join(HERE, "tiny_qwen3_moe.npz"), **tiny_moe_case())
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
