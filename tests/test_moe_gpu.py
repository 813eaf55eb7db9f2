"""GPU: expert-indexed QuantizedLinear (gbx_gather_qlinear_forward / grouped per-expert GEMM) against the oracle, and the
Qwen3-MoE graph on the GPU against the fixture the CPU oracle produced (tests/golden/make_golden.py, case vi)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _experts(oracle, E, N, K, bits, g, seed, dt=torch.bfloat16):
    rng = np.random.default_rng(seed)
    qw = rng.integers(0, 2**32, (E, N, K * bits // 32), dtype=np.uint32)
    qmax = float((1 << bits) - 1)
    s = ((rng.random((E, N, K // g)) + 0.5) * (2.0 / qmax) * 0.02).astype(np.float32)
    z = (-s * qmax / 2.0).astype(np.float32)
    st, zt = torch.from_numpy(s).to(dt), torch.from_numpy(z).to(dt)
    bias = torch.from_numpy((rng.standard_normal((E, N)) * 0.01).astype(np.float32)).to(dt)
    return qw, st, zt, bias


@pytest.mark.parametrize("bits,g", [(2, 128), (3, 32), (4, 64)])
@pytest.mark.parametrize("shared", [True, False])
def test_gather_gemv_matches_oracle(oracle, bits, g, shared):
    from gbx_lm_amd import ops
    E, N, K, k, T = 6, 200, 1024, 4, 5
    qw, s, z, bias = _experts(oracle, E, N, K, bits, g, seed=bits + 10 * shared)
    idx = torch.randint(0, E, (1, T, k))
    x = torch.randn((1, T, 1 if shared else k, 1, K)).to(torch.bfloat16)
    y = ops.gather_quantized_matmul(x.to(DEV), torch.from_numpy(qw.view(np.int32)).view(torch.uint32).to(DEV), s.to(DEV), z.to(DEV),
                                    idx.to(DEV), group_size=g, bits=bits, bias=bias.to(DEV))
    assert y.shape == (1, T, k, 1, N)
    y64 = oracle.gather_qlinear_ref(x, qw, s, z, idx.numpy(), bits, g, bias=bias, acc=np.float64)
    oracle.assert_close(y.reshape(-1, N), y64.reshape(-1, N), "bf16", f"gather b{bits} g{g} shared={shared}")
    # out-of-range expert ids are clamped, not dereferenced
    bad = idx.clone(); bad[0, 0, 0] = E + 7
    y2 = ops.gather_quantized_matmul(x.to(DEV), torch.from_numpy(qw.view(np.int32)).view(torch.uint32).to(DEV), s.to(DEV), z.to(DEV),
                                     bad.to(DEV), group_size=g, bits=bits)
    assert torch.isfinite(y2.float()).all()


def test_gather_grouped_path_many_tokens(oracle):
    """More than 64 (token, slot) pairs: tokens are grouped by expert and each group runs the batched MFMA GEMM."""
    from gbx_lm_amd import ops
    E, N, K, k, T, bits, g = 8, 136, 512, 2, 70, 4, 64
    qw, s, z, _ = _experts(oracle, E, N, K, bits, g, seed=5)
    idx = torch.randint(0, E, (T, k))
    x = torch.randn((T, 1, 1, K)).to(torch.bfloat16)
    y = ops.gather_quantized_matmul(x.to(DEV), torch.from_numpy(qw.view(np.int32)).view(torch.uint32).to(DEV), s.to(DEV), z.to(DEV),
                                    idx.to(DEV), group_size=g, bits=bits)
    y64 = oracle.gather_qlinear_ref(x, qw, s, z, idx.numpy(), bits, g, acc=np.float64)
    oracle.assert_close(y.reshape(-1, N), y64.reshape(-1, N), "bf16", "grouped gather")


def test_qwen3_moe_model_matches_fixture(golden_dir):
    sys.path.insert(0, golden_dir)
    from make_golden import moe_strategy
    from gbx_lm_amd import synthetic as S, utils as U
    from gbx_lm_amd.models.cache import make_prompt_cache
    z = np.load(os.path.join(golden_dir, "tiny_qwen3_moe.npz"))
    model = S.build_model(S.TINY_MOE, moe_strategy(), device="cpu", seed=11).to(DEV)     # same CPU RNG stream as the fixture
    for blk in model.model.layers:
        blk.self_attn.rope.to(DEV)
    ids = z["ids"]
    cache = make_prompt_cache(model)
    with torch.no_grad():
        logits = model(torch.tensor(z["prompt"], device=DEV)[None], cache=cache)[:, -1, :]     # prefill: grouped / GEMM path
        for t in range(8):                                                                        # decode: gather GEMV path
            lp = (logits - torch.logsumexp(logits.float(), -1, keepdim=True).to(logits.dtype)).float().cpu().numpy()[0]
            assert np.abs(lp - z["logprobs"][t]).max() < 0.15, (t, np.abs(lp - z["logprobs"][t]).max())
            if z["gap"][t] > 0.3:
                assert int(np.argmax(lp)) == int(ids[t])
            logits = model(torch.tensor([[ids[t]]], device=DEV), cache=cache)[:, -1, :]
    # the reference API end to end on the MoE family
    toks = [t for t, _, _ in U.generate_step(torch.tensor(z["prompt"]), model, max_tokens=4)]
    assert len(toks) == 4


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_moe_route_and_combine_kernels(dt):
    """gbx_moe_route / gbx_moe_combine against the op-by-op torch formulation of the two routing flavours
    (qqwen3_moe.py:126-141: softmax -> top-k -> renormalise; qmixtral: top-k -> softmax), plus the weighted sum."""
    from gbx_lm_amd import ops
    g = torch.Generator().manual_seed(5)
    for (T, E, k, mode, renorm) in ((1, 128, 8, 0, True), (5, 8, 2, 1, False), (3, 256, 6, 0, False), (2, 60, 4, 0, True)):
        logits = (2.5 * torch.randn(T, E, generator=g)).to(dt)
        inds, scores = ops.moe_route(logits.to(DEV), k, mode, renorm)
        assert inds.dtype == torch.int32 and inds.shape == (T, k) and scores.dtype == dt
        if mode == 0:
            gates = torch.softmax(logits.float(), -1).to(dt)
            ws, wi = torch.topk(gates, k, dim=-1)
            if renorm:
                ws = ws / ws.sum(-1, keepdim=True)
        else:
            top, wi = torch.topk(logits, k, dim=-1)
            ws = torch.softmax(top.float(), -1).to(dt)
        # same SET of experts (ties in the rounded probabilities may order differently), same weights per expert
        for t in range(T):
            got = {int(e): float(w) for e, w in zip(inds[t].cpu(), scores[t].float().cpu())}
            want = {int(e): float(w) for e, w in zip(wi[t], ws[t].float())}
            vals_g, vals_w = sorted(got.values()), sorted(want.values())
            assert np.allclose(vals_g, vals_w, rtol=2e-2, atol=1e-4), (T, E, k, mode)
            common = set(got) & set(want)
            assert len(common) >= k - 1                               # at most one boundary tie may pick another expert
            for e in common:
                assert abs(got[e] - want[e]) <= 2e-2 * abs(want[e]) + 1e-4
        D = 256
        y = torch.randn(T, k, D, generator=g).to(dt)
        out = ops.moe_combine(y.to(DEV), scores)
        ref = (y * scores.cpu().unsqueeze(-1)).sum(dim=-2)
        assert torch.allclose(out.float().cpu(), ref.float(), rtol=2e-2, atol=2e-2)
        res = torch.randn(T, D, generator=g).to(dt)                   # the decoder layer's `h + mlp(...)` fused in
        out2 = ops.moe_combine(y.to(DEV), scores, res.to(DEV))
        assert torch.allclose(out2.float().cpu(), (res + ref).float(), rtol=2e-2, atol=3e-2)


def test_gather_swiglu_launch_equals_the_three_ops():
    """gbx_gather_swiglu_forward (gate, up and SwiGLU of the selected experts in one launch) against
    gate_proj / up_proj / activation run separately through the same library."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.switch_layers import QuantizedSwitchLinear, SwiGLU
    torch.manual_seed(2)
    E, K, N, k = 8, 1024, 384, 3
    def mk():
        m = QuantizedSwitchLinear(K, N, E, bias=False, group_size=128, bits=2)
        m.qweight = torch.randint(0, 2**31 - 1, (E, N, K // 16), dtype=torch.int32).view(torch.uint32).to(DEV)
        m.scales = (torch.rand(E, N, K // 128) * 0.01 + 0.002).to(torch.bfloat16).to(DEV)
        m.zeros = (-m.scales.float() * 1.5).to(torch.bfloat16)
        return m
    gate, up = mk(), mk()
    x = torch.randn(2, 1, 1, K).to(torch.bfloat16).to(DEV)
    inds = torch.tensor([[1, 7, 3], [0, 3, 5]], dtype=torch.int32, device=DEV)
    got = ops.gather_swiglu(x, gate, up, inds)
    assert got is not None and got.shape == (2, k, 1, N)
    want = SwiGLU()(up(x, inds).contiguous(), gate(x, inds).contiguous())
    assert torch.allclose(got.float(), want.float(), rtol=2e-2, atol=2e-3 * want.float().abs().max().item())
