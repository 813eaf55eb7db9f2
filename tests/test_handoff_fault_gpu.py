"""GPU: a fused decode launch whose in-launch hand-off does not arrive must fail IN-BAND and the product loop must recover.

The fused launches (gbx_dattn_forward / gbx_dmlp_forward / gbx_dblock_forward) hand activations between workgroups inside one launch
(reference graph: gbx_lm/models/qllama.py:68-141, decode loop gbx_lm/utils.py:217-338).  A hand-off can fail to arrive -- another tenant
holds CUs, a CU mask, a placement that breaks the kv-group-per-XCD bijection -- and then (a) the launch gives up after
GBX_DATTN_SPIN_LOG2 polls, raises its status word AND the non-finite marker of its accumulator, so the stream, the logits and the token's
logprob are NaN; (b) generate_step reads the status words next to every token, drops the runner, keeps the model on the five-launch block
and REPLAYS the step: the caller sees exactly the tokens of GBX_DATTN=0.  The time-out is forced here by a spin limit of two polls
(each process reads the switch once, hence the subprocesses)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_MODEL = (
    "import sys, warnings, torch; sys.path.insert(0, %r)\n"
    "import bench\n"
    "from gbx_lm_amd import synthetic as S, utils\n"
    "cfg = dict(S.LLAMA3_8B, num_hidden_layers=3, vocab_size=4096)\n"
    "strat = S.make_strategy('S22', 3, 128)\n"
    "for li, bits in enumerate((2, 4, 2)):\n"
    "    for p in strat['model.layers.%%d' %% li]:\n"
    "        strat['model.layers.%%d' %% li][p] = {'bits': [bits], 'group_size': {str(bits): 128}}\n"
    "model = S.build_model(cfg, strat, device='cuda:0', seed=11)\n" % ROOT)


def _run(code, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", _MODEL + code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l[:1].isupper() and " " in l}, out.stderr


_GEN = (
    "prompt = torch.tensor([5, 77, 1030, 9, 400, 3001, 12], dtype=torch.long)\n"
    "with warnings.catch_warnings(record=True) as w:\n"
    "    warnings.simplefilter('always')\n"
    "    toks = [t for t, lp, _ in utils.generate_step(prompt, model, max_tokens=24)]\n"
    "print('TOKENS', ','.join(str(t) for t in toks))\n"
    "print('FELLBACK', int(bool(getattr(model, '_fused_handoff_disabled', False))), sum('five-launch block' in str(x.message) for x in w))\n"
    "r = model._decode_runner\n"
    "print('FUSED', int(any(d is not None for d in getattr(r, 'da', []))))\n")


def test_generate_replays_a_faulted_step_on_the_five_launch_block():
    quiet, _ = _run(_GEN, {"GBX_DATTN": "0"})
    fused, _ = _run(_GEN, {})
    forced, _ = _run(_GEN, {"GBX_DATTN_SPIN_LOG2": "0"})
    assert quiet["FUSED"] == ["0"] and fused["FUSED"] == ["1"], "the launch paths under test"
    assert fused["FELLBACK"][0] == "0", "no time-out on a quiet GPU"
    assert forced["FELLBACK"] == ["1", "1"], "two polls per hand-off must time out, fall back, and warn exactly once"
    assert forced["FUSED"] == ["0"], "the model stays on the five-launch block after a fault"
    assert forced["TOKENS"] == quiet["TOKENS"], "the replayed steps are the five-launch block's"
    assert len(quiet["TOKENS"][0].split(",")) == 24


_INBAND = (
    "from gbx_lm_amd.decode import FusedDecodeRunner\n"
    "caches = bench.make_caches(model, 1, 256, 'cuda:0')\n"
    "r = FusedDecodeRunner(model, caches, use_graph=True)\n"
    "lg = r.step(torch.tensor([7], device='cuda:0'))\n"
    "torch.cuda.synchronize()\n"
    "words = [int(w.item()) for w in r.fault_words()]\n"
    "print('WORDS', ','.join(map(str, words)))\n"
    "print('NANLOGITS', int(torch.isnan(lg.float()).all()))\n"
    "print('NANSTREAM', int(torch.isnan(r.hb.float()).all()))\n"
    "try:\n"
    "    r.dattn_check(); print('CHECK', 'passed')\n"
    "except RuntimeError: print('CHECK', 'raised')\n")


def test_a_timed_out_launch_poisons_its_stream_in_band():
    forced, _ = _run(_INBAND, {"GBX_DATTN_SPIN_LOG2": "0"})
    assert any(int(w) for w in forced["WORDS"][0].split(",")), "status word raised"
    assert forced["NANLOGITS"] == ["1"] and forced["NANSTREAM"] == ["1"], "the residual stream and every logit are NaN"
    assert forced["CHECK"] == ["raised"]
    quiet, _ = _run(_INBAND, {})
    assert not any(int(w) for w in quiet["WORDS"][0].split(",")) and quiet["NANLOGITS"] == ["0"] and quiet["CHECK"] == ["passed"]


def test_fused_launches_beside_a_second_stream_that_holds_cus():
    """The fused launches need their 256 workgroups resident together; a GEMM loop on a second stream takes CUs away for stretches of
    the step.  Allowed outcomes per step: the bit pattern of the quiet run (fixed-point partial sums: order-independent), or a raised status
    word WITH a NaN stream -- never a silent difference."""
    import bench
    from gbx_lm_amd import synthetic as S
    from gbx_lm_amd.decode import FusedDecodeRunner
    cfg = dict(S.LLAMA3_8B, num_hidden_layers=2, vocab_size=2048)
    strat = S.make_strategy("S22", 2, 128)
    for li in range(2):
        for p in strat[f"model.layers.{li}"]:
            strat[f"model.layers.{li}"][p] = {"bits": [2], "group_size": {"2": 128}}
    model = S.build_model(cfg, strat, device=DEV, seed=3)

    def run(busy):
        caches = bench.make_caches(model, 1, 256, DEV)
        r = FusedDecodeRunner(model, caches, use_graph=True)
        assert all(d is not None for d in r.da)
        outs = []
        side = torch.cuda.Stream(device=DEV)
        a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
        b = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
        for t in range(24):
            if busy:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        a = (a @ b).clamp_(-1, 1)
            r.step(torch.tensor([(13 * t + 1) % 2048], device=DEV))
            outs.append(r.hb.clone())
        torch.cuda.synchronize()
        words = [int(w.item()) for w in r.fault_words()]
        return outs, words

    quiet, w0 = run(False)
    assert not any(w0)
    busy, w1 = run(True)
    if any(w1):
        assert torch.isnan(busy[-1].float()).all(), "a raised status word goes with a poisoned stream"
    else:
        for t, (u, v) in enumerate(zip(quiet, busy)):
            assert torch.equal(u, v), f"step {t}: silent difference beside a busy second stream"
