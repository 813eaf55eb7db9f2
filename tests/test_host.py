"""Host-side mirror of the reference interface, on CPU: module shapes, strategy dispatch, loader contract and the
generate loop semantics.  The QuantizedLinear op itself is routed through the oracle here (fixture cpu_oracle_op),
which is test-only plumbing -- the product path has no CPU fallback (test_abi.py)."""
import json
import os

import numpy as np
import pytest
import torch

from gbx_lm_amd import synthetic as S
from gbx_lm_amd import utils as U
from gbx_lm_amd.models import qllama
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear


def test_module_signature_and_shapes():
    m = QuantizedLinear(4096, 1024, True, 128, 3)           # (input_dims, output_dims, bias, group_size, bits)
    assert (m.input_dims, m.output_dims, m.group_size, m.bits) == (4096, 1024, 128, 3)
    assert m.qweight.shape == (1024, 4096 // 32 * 3) and m.qweight.dtype == torch.uint32
    assert m.scales.shape == m.zeros.shape == (1024, 32) and m.scales.dtype == torch.float16
    assert m.channel_scale.shape == (1, 1, 4096) and m.bias.shape == (1024,)
    assert all(not p.requires_grad for p in m.parameters())  # frozen (reference :57-58)
    d = QuantizedLinear(256, 8)                              # defaults: bias False, g64, 4-bit
    assert (d.bias, d.group_size, d.bits) == (None, 64, 4)
    m.init_params(True, True)                                # double-quant + q_perm allocation (:74-108)
    assert m.qstatistic.shape == (32, 32, 32) and m.qstatistic.dtype == torch.uint8
    assert m.qzeros_scales.shape == (32, 32, 1) and m.q_perm.dtype == torch.int16 and m.scales is None


def test_reinit_module_strategy_table():
    """Expected {module: (bits, g, qweight.shape, scales.shape)} for a hand-written strategy."""
    model = S.build_model(S.TINY)
    strat = S.make_strategy("S22", 2)
    strat["model.layers.1"]["q_proj"] = {"bits": [3], "group_size": {"3": 32}}
    strat["model.layers.1"]["down_proj"] = {"bits": [2], "group_size": {"2": 64}}
    QuantizedLinear.reinit_module(model, group_size=64, bits=4, strategy=strat)
    got = {n: (m.bits, m.group_size, tuple(m.qweight.shape), tuple(m.scales.shape))
           for n, m in model.named_modules() if isinstance(m, QuantizedLinear)}
    assert got["model.layers.0.self_attn.q_proj"] == (4, 128, (256, 32), (256, 2))
    assert got["model.layers.1.self_attn.q_proj"] == (3, 32, (256, 24), (256, 8))
    assert got["model.layers.1.mlp.down_proj"] == (2, 64, (256, 32), (256, 8))
    assert got["model.layers.1.mlp.gate_proj"] == (4, 128, (512, 32), (512, 2))
    # uniform path (strategy None) applies to every module
    QuantizedLinear.reinit_module(model, group_size=32, bits=2, strategy=None)
    assert all((m.bits, m.group_size) == (2, 32) for m in model.modules() if isinstance(m, QuantizedLinear))
    bad = S.make_strategy("S40", 2)
    bad["model.layers.0"]["q_proj"]["group_size"]["4"] = 256
    with pytest.raises(AssertionError, match="must be 32, 64 or 128"):
        QuantizedLinear.reinit_module(model, strategy=bad)


def test_strategy_lookup_matches_oracle(oracle):
    model = S.build_model(S.TINY)
    strat = S.make_strategy("S22", 2)
    strat["model.layers.0"]["v_proj"] = {"bits": [2], "group_size": {"2": 32}}
    QuantizedLinear.reinit_module(model, strategy=strat)
    for n, m in model.named_modules():
        if isinstance(m, QuantizedLinear):
            assert (m.bits, m.group_size) == oracle.strategy_lookup(n, strat)


def test_double_quant_module_matches_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "double_quant.npz"))
    K, N, g = int(z["K"]), int(z["N"]), int(z["g"])
    m = QuantizedLinear(K, N, False, g, 4)
    m.init_params(True, False)
    for nm in ("qstatistic", "qzeros_zeros", "qzeros_scales", "qscales_zeros", "qscales_scales"):
        m._buffers[nm] = torch.from_numpy(z[nm])
    m.create_scales_zeros()
    np.testing.assert_array_equal(m.scales.numpy(), z["scales"])
    np.testing.assert_array_equal(m.zeros.numpy(), z["zeros"])
    assert m.qstatistic is None
    before = m.zeros.clone()
    m.set_bias_and_weight()                                   # GBA-native sign -> runtime sign (:151-155)
    assert torch.equal(m.zeros, -before)


def test_load_contract(tmp_path):
    strat = S.make_strategy("S22", 2)
    extra = {"model.layers.0.self_attn.rotary_emb.inv_freq": torch.ones(4),     # dropped by sanitize()
             "model.layers.0.self_attn.q_proj.q_perm": torch.zeros((1, 1, 256), dtype=torch.int16),
             "some.unknown.tensor": torch.zeros(3)}                              # strict=False: ignored
    src = S.write_checkpoint(str(tmp_path), S.TINY, strat, seed=5, extra_tensors=extra)
    model, cfg = U.load_model(tmp_path, device="cpu")
    assert cfg["quantization"] == {"group_size": 64, "bits": 4}
    a, b = S.state_dict_for_disk(src), S.state_dict_for_disk(model)
    assert set(k for k in a if "q_perm" not in k) <= set(b)
    for k in a:
        ta, tb = a[k], b[k]
        if ta.dtype == torch.uint32:
            ta, tb = ta.view(torch.int32), tb.view(torch.int32)
        assert torch.equal(ta, tb), k
    ql = model.model.layers[0].self_attn.q_proj
    assert ql.scales.dtype == torch.bfloat16 and ql.zeros.dtype == torch.bfloat16   # utils.py:841-843
    assert ql.channel_scale.dtype == torch.bfloat16                                  # "scale" in key -> bf16 too
    assert (ql.bits, ql.group_size) == (4, 128)
    # missing safetensors / config raise like the reference
    with pytest.raises(FileNotFoundError):
        U.load_model(tmp_path / "nope", device="cpu")
    os.remove(tmp_path / "model.safetensors")
    with pytest.raises(FileNotFoundError):
        U.load_model(tmp_path, device="cpu")


def test_load_rejects_bad_group_and_shape(tmp_path):
    S.write_checkpoint(str(tmp_path), S.TINY, None, seed=1, quantization={"group_size": 64, "bits": 4})
    cfg = json.load(open(tmp_path / "config.json"))
    cfg["quantization"]["group_size"] = 48
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    with pytest.raises(AssertionError, match="must be 32, 64 or 128"):
        U.load_model(tmp_path, device="cpu")
    cfg["quantization"] = {"group_size": 64, "bits": 2}      # checkpoint tensors are 4-bit shaped
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    with pytest.raises(ValueError, match="shape mismatch"):
        U.load_model(tmp_path, device="cpu")


def test_generate_loop_semantics(tmp_path, cpu_oracle_op):
    strat = S.make_strategy("S22", 2)
    S.write_checkpoint(str(tmp_path), S.TINY, strat, seed=3)
    model, tok = U.load(str(tmp_path))
    assert isinstance(model, qllama.Model) and tok.eos_token_ids == {511}
    prompt = [5, 17, 42, 7, 99]
    steps = list(U.generate_step(torch.tensor(prompt), model, max_tokens=6))
    assert len(steps) == 6
    t0, lp0, hs0 = steps[0]
    assert isinstance(t0, int) and lp0.shape == (512,) and hs0 is None
    assert abs(torch.logsumexp(lp0.float(), -1).item()) < 2e-2           # logprobs = logits - logsumexp
    assert t0 == int(torch.argmax(lp0))                                   # greedy default sampler
    # chunked prefill gives the same first token as one-shot prefill
    s2 = list(U.generate_step(torch.tensor(prompt), model, max_tokens=2, prefill_step_size=2))
    assert s2[0][0] == t0
    # incremental decode == full re-forward (KV cache append + causal mask with offset)
    ids = [s[0] for s in steps]
    full = model(torch.tensor([prompt + ids[:3]]))[0, -1]
    assert int(torch.argmax(full)) == ids[3]
    with pytest.raises(ValueError, match="Wrong number of layers"):
        next(U.generate_step(torch.tensor(prompt), model, prompt_cache=[None]))
    with pytest.raises(NotImplementedError):
        next(U.generate_step(torch.tensor(prompt), model, kv_bits=4))
    # stream_generate / generate: response fields, finish reasons, EOS stop
    rs = list(U.stream_generate(model, tok, prompt, max_tokens=4))
    assert rs[-1].finish_reason == "length" and rs[-1].generation_tokens == 4 and rs[-1].prompt_tokens == 5
    assert [r.token for r in rs[:-1]] == ids[:4] and all(r.finish_reason is None for r in rs[:-1])
    tok.add_eos_token(ids[2])
    rs = list(U.stream_generate(model, tok, prompt, max_tokens=6))
    assert rs[-1].finish_reason == "stop" and rs[-1].token == ids[2] and len(rs) == 3
    text = U.generate(model, tok, "hello world", max_tokens=3)
    assert isinstance(text, str)


def test_eminf_optimize_matches_oracle(golden_dir, oracle):
    from gbx_lm_amd.infer_opt import eminf_optimize
    z = np.load(os.path.join(golden_dir, "eminf.npz"))
    best, h = eminf_optimize(torch.from_numpy(z["logits"]))
    np.testing.assert_allclose(best.numpy(), z["best"], rtol=0, atol=2e-4)
    assert abs(float(h) - float(z["entropy"])) < 2e-4
    assert float(h) < float(oracle.eminf_optimize(z["logits"], num_steps=0)[1])      # entropy went down
    b2, h2 = eminf_optimize(torch.from_numpy(z["sharp"]))
    np.testing.assert_array_equal(b2.numpy(), z["sharp_best"])                        # below target: untouched
    assert abs(float(h2) - float(z["sharp_entropy"])) < 1e-6


def test_eminf_generate_loop(tmp_path, cpu_oracle_op):
    from gbx_lm_amd.infer_opt import eminf_generate
    S.write_checkpoint(str(tmp_path), S.TINY, S.make_strategy("S22", 2), seed=3)
    model, tok = U.load(str(tmp_path))
    ids = eminf_generate(model, tok, [5, 17, 42], max_tokens=5, rng=np.random.default_rng(0))
    assert 1 <= len(ids) <= 5 and all(0 <= t < 512 for t in ids)
    ids2 = eminf_generate(model, tok, [5, 17, 42], max_tokens=5, rng=np.random.default_rng(0))
    assert ids == ids2                                                                # seeded numpy sampling
    with pytest.raises(NotImplementedError):
        eminf_generate(model, tok, [1], max_tokens=1, max_kv_size=64)


def test_eminf_generate_step_and_prompt_cache(tmp_path, cpu_oracle_op, capsys):
    """infer_opt.py:10-28,168-245: the generator form yields (token, logprob of that token, None) and draws the same
    tokens as eminf_generate under the same numpy stream; with a PromptCache the second request of a conversation feeds
    only its new tokens and still reproduces the from-scratch tokens."""
    from gbx_lm_amd.infer_opt import eminf_generate, eminf_generate_step, get_input_ids
    from gbx_lm_amd.prompt_cache import PromptCache
    from gbx_lm_amd.models.cache import make_prompt_cache
    S.write_checkpoint(str(tmp_path), S.TINY, S.make_strategy("S22", 2), seed=3)
    model, tok = U.load(str(tmp_path))
    ids = eminf_generate(model, tok, [5, 17, 42], max_tokens=6, rng=np.random.default_rng(1))
    steps = list(eminf_generate_step(model, tok, [5, 17, 42], max_tokens=6, rng=np.random.default_rng(1)))
    assert [t for t, _, _ in steps] == [t for t in ids if t not in tok.eos_token_ids][:len(steps)] and len(steps) >= 1
    for t, lp, hs in steps:
        assert hs is None and lp.ndim == 0 and float(lp) <= 0.0
    # get_input_ids: no cache / use_cache=False -> fresh cache + the whole prompt
    todo, hit, cache = get_input_ids(None, model, [1, 2, 3], [1, 2], "k")
    assert todo == [1, 2, 3] and not hit and cache[0].offset == 0
    pc = PromptCache()
    todo, hit, cache = get_input_ids(pc, model, [1, 2, 3], [1, 2], "k", use_cache=False)
    assert todo == [1, 2, 3] and not hit and pc.cache is None
    # a conversation through the prompt cache: turn 1 (miss), turn 2 extends it (hit: only the new tokens are fed)
    conv1, conv2 = [5, 17, 42, 9], [5, 17, 42, 9, 33, 34, 35]
    a1 = eminf_generate(model, tok, conv1 + [4], conv1, max_tokens=3, prompt_cache=pc, rng=np.random.default_rng(2))
    assert pc.cache is not None and pc.tokens_no_gen == conv1
    # the cache now also holds the generation prompt and the sampled tokens of turn 1: the reference's own bookkeeping
    # (PromptCache.tokens_no_gen) decides what is reused, so mirror what its caller does and trim to the conversation
    for c in pc.cache:
        c.trim(c.offset - len(conv1))
    capsys.readouterr()
    a2 = eminf_generate(model, tok, conv2 + [4], conv2, max_tokens=4, prompt_cache=pc, rng=np.random.default_rng(3))
    assert "Use input cache" in capsys.readouterr().out
    want = eminf_generate(model, tok, conv2 + [4], conv2, max_tokens=4, rng=np.random.default_rng(3))
    assert a2 == want and len(a1) >= 1
    # identical request again: nothing new to process -> last token re-fed on a cache trimmed by one (no duplicate KV rows)
    for c in pc.cache:
        c.trim(c.offset - len(conv2))
    todo, hit, cache = get_input_ids(pc, model, conv2, conv2, pc.model_key)
    assert hit and todo == conv2[-1:] and cache[0].offset == len(conv2) - 1


def test_load_gba_native_conversion_and_double_quant(tmp_path):
    """GBA-native checkpoints (utils.py:828-838,864-873): qweight/scales/zeros stored transposed, zeros with the
    GBA sign (w = q*s - z), optionally double-quantised statistics instead of scales/zeros."""
    from safetensors.torch import save_file
    strat = S.make_strategy("S40", 2, 64)
    src = S.write_checkpoint(str(tmp_path / "mlx"), S.TINY, strat, seed=8)
    ref, _ = U.load_model(tmp_path / "mlx", device="cpu")
    # --- plain GBA-native layout ---
    sd = {}
    for k, v in S.state_dict_for_disk(src).items():
        if k.endswith("qweight"):
            sd[k] = v.view(torch.int32).t().contiguous()
        elif k.endswith("scales"):
            sd[k] = v.to(torch.float16).t().contiguous()
        elif k.endswith("zeros"):
            sd[k] = (-v.to(torch.float32)).to(torch.float16).t().contiguous()
        else:
            sd[k] = v
    nat = tmp_path / "native"
    os.makedirs(nat)
    save_file(sd, str(nat / "model.safetensors"))
    for f in ("config.json", "quant_strategy.json"):
        open(nat / f, "w").write(open(tmp_path / "mlx" / f).read())
    conv, _ = U.load_model(nat, device="cpu", is_conversion=True)
    a, b = S.state_dict_for_disk(ref), S.state_dict_for_disk(conv)
    for k in a:
        if k.endswith("qweight"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), k
        elif k.endswith(("scales", "zeros")):
            # fp16 round trip of bf16 values is exact for these magnitudes; sign restored by post_processing_and_release
            assert torch.allclose(a[k].float(), b[k].float(), rtol=1e-2, atol=1e-6), k
            assert (torch.sign(a[k].float()) == torch.sign(b[k].float())).all(), k
    # --- double-quantised statistics: scales/zeros replaced by qstatistic + 4 side tensors ---
    q = ref.model.layers[0].mlp.gate_proj
    K, N, g = q.input_dims, q.output_dims, q.group_size
    kg, ng = K // g, N // 32
    gen = torch.Generator().manual_seed(0)
    dq = {k: v for k, v in sd.items() if not (k.endswith("scales") or k.endswith("zeros"))}
    for name, m in ref.named_modules():
        if isinstance(m, QuantizedLinear):
            kgm, ngm = m.input_dims // m.group_size, -(-m.output_dims // 32)
            dq[name + ".qstatistic"] = torch.randint(0, 256, (kgm, ngm, 32), dtype=torch.uint8, generator=gen)
            dq[name + ".qzeros_zeros"] = torch.full((kgm, ngm, 1), 7.5, dtype=torch.float16)
            dq[name + ".qzeros_scales"] = torch.full((kgm, ngm, 1), 0.01, dtype=torch.float16)
            dq[name + ".qscales_zeros"] = torch.full((kgm, ngm, 1), -1.0, dtype=torch.float16)
            dq[name + ".qscales_scales"] = torch.full((kgm, ngm, 1), 0.002, dtype=torch.float16)
    dqd = tmp_path / "dq"
    os.makedirs(dqd)
    save_file(dq, str(dqd / "model.safetensors"))
    for f in ("config.json", "quant_strategy.json"):
        open(dqd / f, "w").write(open(tmp_path / "mlx" / f).read())
    m2, _ = U.load_model(dqd, device="cpu", is_conversion=True)
    g2 = m2.model.layers[0].mlp.gate_proj
    assert g2.qstatistic is None and g2.scales.shape == (N, kg) and g2.zeros.shape == (N, kg)
    st = dq["model.layers.0.mlp.gate_proj.qstatistic"]
    want_s = ((((st & 0xF0) >> 4).to(torch.float16) + 1.0) * 0.002).view(kg, N).t()
    want_z = -(((st & 0x0F).to(torch.float16) - 7.5) * 0.01).view(kg, N).t()          # negated by set_bias_and_weight
    assert torch.allclose(g2.scales.float(), want_s.float(), rtol=1e-2, atol=1e-6)
    assert torch.allclose(g2.zeros.float(), want_z.float(), rtol=1e-2, atol=1e-6)


def _write_native(tmp_path, name, double_quant):
    """A GBA-native twin of a runtime-layout tiny checkpoint: transposed qweight / scales / zeros, GBA-sign zeros in fp16,
    or (double_quant) qstatistic + side tensors instead of scales / zeros."""
    from safetensors.torch import save_file
    strat = S.make_strategy("S22", 2, 64)
    src = S.write_checkpoint(str(tmp_path / "mlx"), S.TINY, strat, seed=8)
    sd = {}
    gen = torch.Generator().manual_seed(0)
    for k, v in S.state_dict_for_disk(src).items():
        if k.endswith("qweight"):
            sd[k] = v.view(torch.int32).t().contiguous()
        elif k.endswith("scales"):
            if not double_quant:
                sd[k] = v.to(torch.float16).t().contiguous()
        elif k.endswith("zeros"):
            if not double_quant:
                sd[k] = (-v.to(torch.float32)).to(torch.float16).t().contiguous()
        else:
            sd[k] = v
    if double_quant:
        for mname, m in src.named_modules():
            if isinstance(m, QuantizedLinear):
                kg, ng = m.input_dims // m.group_size, -(-m.output_dims // 32)
                sd[mname + ".qstatistic"] = torch.randint(0, 256, (kg, ng, 32), dtype=torch.uint8, generator=gen)
                sd[mname + ".qzeros_zeros"] = torch.full((kg, ng, 1), 7.5, dtype=torch.float16)
                sd[mname + ".qzeros_scales"] = torch.full((kg, ng, 1), 0.004, dtype=torch.float16)
                sd[mname + ".qscales_zeros"] = torch.full((kg, ng, 1), -1.0, dtype=torch.float16)
                sd[mname + ".qscales_scales"] = torch.full((kg, ng, 1), 0.0008, dtype=torch.float16)
    nat = tmp_path / name
    os.makedirs(nat)
    save_file(sd, str(nat / "model.safetensors"))
    for f in ("config.json", "quant_strategy.json", "tokenizer.json", "tokenizer_config.json"):
        open(nat / f, "w").write(open(tmp_path / "mlx" / f).read())
    return nat


@pytest.mark.parametrize("double_quant", [False, True])
def test_convert_gba_native_repo(tmp_path, double_quant):
    """gba2mlx.convert (gbx_lm/gba2mlx.py:93-161): GBA-native repo -> runtime layout on disk; plain load() of the result
    holds exactly the tensors load_model(is_conversion=True) builds in memory."""
    from gbx_lm_amd import gba2mlx
    nat = _write_native(tmp_path, "tiny-w4-g64-native", double_quant)
    out = tmp_path / "converted-mlx"
    gba2mlx.convert(str(nat), str(out))
    files = sorted(p.name for p in out.iterdir())
    assert {"config.json", "quant_strategy.json", "model.safetensors", "model.safetensors.index.json", "tokenizer.json"} <= set(files)
    cfg = json.load(open(out / "config.json"))
    assert cfg["quantization"] == {"group_size": 64, "bits": 4} and list(cfg) == sorted(cfg)
    idx = json.load(open(out / "model.safetensors.index.json"))
    assert "model.layers.0.self_attn.q_proj.qweight" in idx["weight_map"] and idx["metadata"]["total_size"] > 0
    direct, _ = U.load_model(nat, device="cpu", is_conversion=True)
    conv, tok = U.load(str(out))
    conv = conv.to("cpu")
    a, b = S.state_dict_for_disk(direct), S.state_dict_for_disk(conv)
    assert set(a) == set(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k].view(torch.int16) if a[k].dtype in (torch.bfloat16, torch.float16) else
                                                        a[k].view(torch.int32) if a[k].dtype == torch.uint32 else a[k],
                                                        b[k].view(torch.int16) if b[k].dtype in (torch.bfloat16, torch.float16) else
                                                        b[k].view(torch.int32) if b[k].dtype == torch.uint32 else b[k]), k
    assert not any("qstatistic" in k for k in b)
    assert gba2mlx.extract_bits_and_group_size("GreenBitAI/Llama-3-8B-w3-g128") == (3, 128)
    assert gba2mlx.extract_bits_and_group_size("plain-name") == (None, None)
    with pytest.raises(NotImplementedError):
        gba2mlx.convert(str(nat), str(out), upload_repo="a/b-mlx")
    big = {f"t{i}": torch.zeros(1 << 20, dtype=torch.float32) for i in range(5)}      # 4 MiB each
    assert [len(s) for s in gba2mlx.make_shards(big, max_file_size_gb=0)] == [0, 1, 1, 1, 1, 1] or len(gba2mlx.make_shards(big)) == 1
