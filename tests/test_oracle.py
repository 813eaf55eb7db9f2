"""Oracle vs the committed golden vectors and vs itself (numpy restatement, formula restatement, C restatement).
CPU only.  Pins the oracle before anything is compared against it."""
import json
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st


def _h(a, dt):
    t = torch.from_numpy(a.view(np.int16))
    return t.view(torch.bfloat16 if dt == "bf16" else torch.float16)


def test_known_answer_rows(oracle, c_oracle, golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "known_answer.json")))
    for b, rec in ka.items():
        b = int(b)
        qw = np.array([[int(w, 16) for w in rec["words"]]], dtype=np.uint32)
        want = np.array([rec["fields"]], dtype=np.int32)
        assert (oracle.unpack(qw, b) == want).all()
        assert (oracle.unpack_formula(qw, b) == want).all()
        assert (c_oracle.unpack(qw, b) == want).all()
        assert (oracle.pack(want, b) == qw).all()


def test_three_bit_straddlers(oracle):
    # fields 10 and 21 cross word boundaries; set only them
    q = np.zeros((1, 32), dtype=np.int32)
    q[0, 10], q[0, 21] = 0b101, 0b110
    qw = oracle.pack(q, 3)
    assert qw[0, 0] >> 30 == 0b01 and qw[0, 1] & 1 == 1          # 101 -> low two bits in word0, high bit in word1
    assert qw[0, 1] >> 31 == 0 and qw[0, 2] & 3 == 0b11          # 110 -> bit0 in word1[31], bits 1..2 in word2
    assert (oracle.unpack(qw, 3) == q).all()


@settings(max_examples=60, deadline=None)
@given(bits=st.sampled_from([2, 3, 4]), n=st.integers(1, 5), kmul=st.integers(1, 6), seed=st.integers(0, 2**31 - 1))
def test_pack_unpack_roundtrip(bits, n, kmul, seed):
    from oracle import gba_oracle as O, c_oracle as C
    rng = np.random.default_rng(seed)
    k = 32 * kmul
    q = rng.integers(0, 1 << bits, size=(n, k)).astype(np.int32)
    qw = O.pack(q, bits)
    assert qw.shape == (n, k * bits // 32)
    assert (O.unpack(qw, bits) == q).all()
    assert (O.unpack_formula(qw, bits) == q).all()
    assert (C.unpack(qw, bits) == q).all()
    words = rng.integers(0, 2**32, size=(n, k * bits // 32), dtype=np.uint64).astype(np.uint32)
    assert (O.pack(O.unpack(words, bits), bits) == words).all()


def test_small_cases_match_golden(oracle, c_oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "qlinear_small.npz"))
    for b in (2, 3, 4):
        for g in (32, 64, 128):
            for dt in ("bf16", "f16"):
                key = f"b{b}_g{g}_{dt}"
                qw = z[key + "_qweight"]
                sc, ze, x = _h(z[key + "_scales"], dt), _h(z[key + "_zeros"], dt), _h(z[key + "_x"], dt)
                bias = _h(z[key + "_bias"], dt) if key + "_bias" in z.files else None
                y64 = z[key + "_y64"]
                assert (oracle.unpack(qw, b) == z[key + "_q"]).all()
                got64 = oracle.qlinear_ref(x, qw, sc, ze, b, g, bias=bias, acc=np.float64)
                np.testing.assert_array_equal(got64, y64)
                c64 = c_oracle.qlinear(x, qw, sc, ze, b, g, bias=bias, acc64=True)
                np.testing.assert_allclose(c64, y64, rtol=0, atol=1e-12)
                rms = np.sqrt(np.mean(y64 ** 2))
                y32 = oracle.qlinear_ref(x, qw, sc, ze, b, g, bias=bias)
                assert np.abs(y32 - y64).max() <= 1e-4 * rms      # fp32 oracle sits within 1e-4 rms of fp64
                c32 = c_oracle.qlinear(x, qw, sc, ze, b, g, bias=bias, acc64=False)
                assert np.abs(c32 - y64).max() <= 1e-4 * rms
                w_np = oracle.dequant(qw, sc, ze, b, g)
                w_c = c_oracle.dequant(qw, sc, ze, b, g)
                np.testing.assert_array_equal(w_np, w_c)          # one fp32 mul + one fp32 add, both sides


def test_double_quant_expansion(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "double_quant.npz"))
    sc, ze = oracle.create_scales_zeros(z["qstatistic"], z["qzeros_zeros"], z["qzeros_scales"], z["qscales_zeros"],
                                        z["qscales_scales"], int(z["K"]), int(z["N"]), int(z["g"]))
    assert sc.shape == (int(z["N"]), int(z["K"]) // int(z["g"])) and sc.dtype == torch.float16
    np.testing.assert_array_equal(sc.numpy(), z["scales"])
    np.testing.assert_array_equal(ze.numpy(), z["zeros"])
    # nibble split by hand for one element: scales from the high nibble, zeros from the low nibble
    qs = z["qstatistic"]
    hi, lo = (qs[1, 0, 5] & 0xF0) >> 4, qs[1, 0, 5] & 0x0F
    want_s = (np.float16(hi) - z["qscales_zeros"][1, 0, 0]) * z["qscales_scales"][1, 0, 0]
    want_z = (np.float16(lo) - z["qzeros_zeros"][1, 0, 0]) * z["qzeros_scales"][1, 0, 0]
    assert sc[5, 1].item() == np.float16(want_s) and ze[5, 1].item() == np.float16(want_z)


def test_validation_errors(oracle):
    qw = np.zeros((2, 8), dtype=np.uint32)
    s = torch.ones((2, 1), dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        oracle.unpack(qw, 5)
    with pytest.raises(ValueError):
        oracle.dequant(qw, s, s, 4, 48)


def test_strategy_lookup_quirks(oracle):
    strat = {"model.layers.0": {
        "q_proj": {"bits": [2], "group_size": {"2": 32}},
        "v_proj": {"bits": [3], "group_size": {"3": 64}},
        "up_proj": {"bits": [4], "group_size": {"4": 128}},
        "qkv_proj": {"bits": [4], "group_size": {"4": 32}},
    }}
    assert oracle.strategy_lookup("model.layers.0.self_attn.q_proj", strat) == (2, 32)
    # substring quirk: "qkv_proj" contains "v_proj", which is scanned first
    assert oracle.strategy_lookup("model.layers.0.self_attn.qkv_proj", strat) == (3, 64)
    # "gate_up_proj" contains "up_proj" (scanned before gate_up_proj); "gate_proj" is not a substring
    assert oracle.strategy_lookup("model.layers.0.mlp.gate_up_proj", strat) == (4, 128)
    # KeyError fall-through: k_proj missing -> nothing else matches -> error
    with pytest.raises(KeyError):
        oracle.strategy_lookup("model.layers.0.self_attn.k_proj", strat)
    bad = {"model.layers.0": {"q_proj": {"bits": [4], "group_size": {"4": 256}}}}
    with pytest.raises(AssertionError):
        oracle.strategy_lookup("model.layers.0.self_attn.q_proj", bad)
