"""gbx_attn_oproj_decode (attention + o_proj slices in one launch, fixed-point accumulator) and the x_acc / x_out / zero_buf
options of gbx_qlinear_fused_forward, against the oracle (fp64 re-expression of qllama.py:78-96 + the GBA dequant) and
against the two-launch path they replace."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.experimental]
DEV = "cuda:0"


def _mod(oracle, n, k, bits, g, seed, act):
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    d = oracle.synth_layer(n, k, bits, g, seed=seed, act_dtype=act)
    m = QuantizedLinear(k, n, False, g, bits)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m, d


def _attention_ref64(q, k, v, kc, vc, inv_freq, pos, nh, nkv, hd, dt):
    """fp64 attention of the new token on the CPU: RoPE with one rounding to the activation dtype (mx.fast.rope), cache rows
    0..pos-1 plus the new k / v, softmax in fp64, output rounded to the activation dtype like the SDPA output."""
    from gbx_lm_amd.models.rope_utils import apply_rope
    qr = apply_rope(q.view(1, nh, 1, hd), inv_freq, pos, 1.0, False).cpu().double()
    kr = apply_rope(k.view(1, nkv, 1, hd), inv_freq, pos, 1.0, False)
    K = torch.cat([kc[:, :, :pos], kr], 2).cpu().double().repeat_interleave(nh // nkv, 1)
    V = torch.cat([vc[:, :, :pos], v.view(1, nkv, 1, hd)], 2).cpu().double().repeat_interleave(nh // nkv, 1)
    p = torch.softmax((qr @ K.transpose(-1, -2)) * hd ** -0.5, -1)
    return (p @ V).reshape(-1).to(dt), kr


@pytest.mark.parametrize("bits,g", [(2, 128), (2, 64), (3, 128), (4, 128), (4, 64)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("pos", [0, 5, 77, 300])
def test_attn_oproj_matches_oracle_and_two_launch_path(oracle, bits, g, dt, pos):
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.rope_utils import RoPE
    act = "bf16" if dt == torch.bfloat16 else "f16"
    hd, nh, nkv, cap, D = 128, 8, 2, 512, 640
    o_proj, d = _mod(oracle, D, nh * hd, bits, g, 3, act)
    assert ops.attn_oproj_supported(nh, nkv, hd, o_proj)
    q, k, v = (torch.randn(c * hd).to(dt).to(DEV) for c in (nh, nkv, nkv))
    kc = torch.randn(1, nkv, cap, hd).to(dt).to(DEV)
    vc = torch.randn(1, nkv, cap, hd).to(dt).to(DEV)
    kc2, vc2 = kc.clone(), vc.clone()
    rope = RoPE(hd).to(DEV)
    posd = torch.tensor([pos], device=DEV)
    h = torch.randn(D).to(dt).to(DEV)

    acc_all = torch.zeros(D + nkv * ops.ATTN_OPROJ_HANDOFF_WORDS, dtype=torch.int64, device=DEV)
    acc = acc_all[:D]
    ops.attn_oproj_decode(q, k, v, kc, vc, rope.inv_freq, posd, o_proj, acc_all, nh, nkv, hd, cap, 1.0, hd ** -0.5, False)
    y_new = (acc.double() * 2.0 ** -32).cpu().numpy()
    assert bool((acc_all[D:] >> 32 == 1).all())           # every hand-off word was published

    # oracle: fp64 attention, GBA dequant-matmul in fp64 on its (rounded) output
    a64, kr = _attention_ref64(q, k, v, kc2, vc2, rope.inv_freq, pos, nh, nkv, hd, dt)
    y64 = oracle.qlinear_ref(a64.float()[None], d["qweight"], d["scales"], d["zeros"], bits, g, acc=np.float64)[0]
    # the attention output is an activation-dtype tensor: a rounding difference of one unit in one of its 1024 entries moves y
    # by |w| * ulp, so compare at the output's own tolerance
    oracle.assert_close(torch.from_numpy(y_new)[None].to(dt), y64[None], act, "attn + o_proj")
    # cache append: identical to the two-launch path, bit for bit
    out = torch.zeros(nh * hd, dtype=dt, device=DEV)
    ops.attn_decode(q, k, v, kc2, vc2, rope.inv_freq, posd, out, nh, nkv, hd, cap, 1.0, hd ** -0.5, False)
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)
    # ... and so is the attention itself: the o_proj of the two-launch path on ITS attention output agrees to fp32 rounding
    h2 = h.clone()
    ops.qlinear_fused(ops.make_fused_desc(out, (o_proj,), (h2,), residual=h2), out.device)
    want = h.float() + (acc.double() * 2.0 ** -32).float().to(dt).float()
    got_new = want.to(dt)
    diff = (got_new.float() - h2.float()).abs()
    ulp = torch.maximum(h2.float().abs(), torch.tensor(1e-3, device=DEV)) * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10)
    assert (diff <= 2 * ulp).all(), float((diff / ulp).max())
    assert (diff == 0).float().mean() > 0.9          # accumulation order only: almost every entry identical


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n", [1030, 14336])
def test_x_acc_prologue_and_zero_buf(oracle, bits, dt, n):
    """gate|up launch with x = h + round(acc * 2^-32) formed in its prologue == the same launch on the materialised sum;
    x_out receives that sum; the down launch clears the accumulator."""
    from gbx_lm_amd import ops
    act = "bf16" if dt == torch.bfloat16 else "f16"
    k, g = 4096, 128
    gate, _ = _mod(oracle, n, k, bits, g, 1, act)
    up, _ = _mod(oracle, n, k, bits, g, 2, act)
    pgu = ops.pack_scales_zeros(gate, up)
    h = torch.randn(k).to(dt).to(DEV)
    r = (torch.randn(k, device=DEV) * 0.7)
    acc = torch.round(r.double() * 2.0 ** 32).to(torch.int64)
    nw = (1 + 0.1 * torch.randn(k)).to(dt).to(DEV)
    hsum = (h.float() + (acc.double() * 2.0 ** -32).float().to(dt).float()).to(dt)
    a_ref = torch.zeros(n, dtype=dt, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(hsum, (gate, up), (a_ref,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pgu), h.device)
    a_new = torch.zeros(n, dtype=dt, device=DEV)
    x_out = torch.zeros(k, dtype=dt, device=DEV)
    ops.qlinear_fused(ops.make_fused_desc(h, (gate, up), (a_new,), norm_weight=nw, eps=1e-5, swiglu=True, packed=pgu,
                                          x_acc=acc, x_out=x_out), h.device)
    assert torch.equal(x_out, hsum)
    # same arithmetic on the same vector; the two instantiations may differ in the last bit of the RMS statistic: one ulp on a few
    # normalised activations, which the SwiGLU epilogue's three roundings (gate, silu(gate), product) can turn into up to three ulps
    # of an output -- on a handful of entries; most are identical
    # (an output far below the typical magnitude -- a cancelling up row, a strongly negative gate -- inherits the ABSOLUTE wobble of
    # its pre-activations: units are ulps of max(|a|, mean |a|))
    d = (a_new.float() - a_ref.float()).abs()
    ulp = torch.maximum(a_ref.float().abs(), a_ref.float().abs().mean()) * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10)
    assert (d <= 3.01 * ulp).all(), float((d / ulp).max())
    assert (d <= 1.01 * ulp).float().mean() > 0.99 and (d == 0).float().mean() > 0.8
    # zero_buf on a residual launch
    dn, _ = _mod(oracle, 256, n if n % 128 == 0 else 1024, bits, g, 5, act)
    xa = torch.randn(dn.input_dims).to(dt).to(DEV)
    y1, y2 = torch.zeros(256, dtype=dt, device=DEV), torch.zeros(256, dtype=dt, device=DEV)
    pdn = ops.pack_scales_zeros(dn)
    ops.qlinear_fused(ops.make_fused_desc(xa, (dn,), (y1,), residual=x_out[:256].contiguous(), packed=pdn), xa.device)
    ops.qlinear_fused(ops.make_fused_desc(xa, (dn,), (y2,), residual=x_out[:256].contiguous(), packed=pdn, zero_buf=acc), xa.device)
    assert torch.equal(y1, y2)
    assert int(acc.abs().sum().item()) == 0


def test_x_acc_rejections(oracle):
    from gbx_lm_amd import ops, _lib
    gate, _ = _mod(oracle, 64, 4096, 2, 128, 1, "bf16")
    h = torch.randn(4096).to(torch.bfloat16).to(DEV)
    y = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
    acc = torch.zeros(4096, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError):        # no packed companion
        ops.qlinear_fused(ops.make_fused_desc(h, (gate,), (y,), x_acc=acc), h.device)
    with pytest.raises(ValueError):          # x_out aliasing x
        ops.qlinear_fused(ops.make_fused_desc(h, (gate,), (y,), packed=ops.pack_scales_zeros(gate), x_acc=acc, x_out=h), h.device)
    assert not _lib.exp_lib().gbx_attn_oproj_supported(32, 8, 64, 4096, 2, 128)
    assert not _lib.exp_lib().gbx_attn_oproj_supported(24, 8, 128, 3072, 4, 128)
    assert _lib.exp_lib().gbx_attn_oproj_supported(32, 8, 128, 4096, 2, 128)


def test_runner_four_launch_block_matches_five_launch_block():
    """FusedDecodeRunner with the attention + o_proj launch (short cache) against GBX_FUSE_AO=0 on the same weights: same greedy
    ids over 24 tokens and logits within the accumulation-order tolerance."""
    import os
    from gbx_lm_amd import synthetic as S
    from gbx_lm_amd.decode import FusedDecodeRunner
    from gbx_lm_amd.models.cache import KVCache
    cfg = dict(S.TINY, hidden_size=1024, intermediate_size=2048, num_attention_heads=8, num_key_value_heads=2, num_hidden_layers=3)
    model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=DEV, seed=3)

    def run(fuse):
        os.environ["GBX_FUSE_AO"] = "1" if fuse else "0"          # opt-in launch structure
        try:
            at = model.model.layers[0].self_attn
            caches = []
            for _ in model.model.layers:
                c = KVCache(step=256)
                like = torch.empty((1, at.n_kv_heads, 1, at.head_dim), dtype=torch.bfloat16, device=DEV)
                c.reserve(1, like, like)
                caches.append(c)
            r = FusedDecodeRunner(model, caches, use_graph=True)
            assert r.fuse_ao == fuse
            ids, logits = [], []
            tok = torch.tensor([5], device=DEV)
            for _ in range(24):
                lg = r.step(tok).clone()
                tok = torch.argmax(lg, -1)
                ids.append(int(tok.item()))
                logits.append(lg.float())
            assert int(r.acc.abs().sum().item()) == 0
            return ids, torch.stack(logits)
        finally:
            os.environ.pop("GBX_FUSE_AO", None)
    ids_a, lg_a = run(True)
    ids_b, lg_b = run(False)
    tol = 0.05 * float(lg_b[0].std()) + 0.02            # accumulation order (fixed point vs fp32) moves a few activations by one ulp
    assert float((lg_a[0] - lg_b[0]).abs().max()) <= tol
    assert ids_a[0] == ids_b[0]
