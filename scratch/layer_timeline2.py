"""Timeline of one decoded token with the two-launch block (round 5: dattn, dmlp), from in-kernel s_memrealtime stamps.
    GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so python scratch/layer_timeline2.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
dev = torch.device("cuda:0")
NL = 32
REG = 4096 * 16
from gbx_lm_amd import synthetic as S0
bits = [S0.strategy_bits("S22", i, NL) for i in range(NL)]
per = [2 if (b == 2 or (b == 3 and i != NL - 1 and os.environ.get('GBX_DMLP_MX3', '1') == '2') or os.environ.get('GBX_DMLP_VEC', '0') != '0') else 3 for i, b in enumerate(bits)]   # (S22: isolated 4-bit blocks stay on three launches)      # (round 6: 3- / 4-bit blocks take the fused MLP launch too)
SLOTS = sum(per)
base = np.cumsum([0] + per)
stamps = torch.zeros(SLOTS * REG, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
os.environ["GBX_STAMP_SLOTS"] = str(SLOTS)
os.environ["GBX_DATTN"] = "1"
os.environ.setdefault("GBX_DMLP", "1")
os.environ["GBX_DBLOCK"] = "0"
if os.environ.get("GBX_PROBE_MODE"):      # round-6 step 0 (ii): an LDS-DMA stream beside dattn's hand-off chain (stamps build only)
    pool = torch.full(((24 * 32 + 40) << 20,), 1, dtype=torch.uint8, device=dev)
    os.environ["GBX_PROBE_SRC"] = str(pool.data_ptr())

from gbx_lm_amd import _lib, synthetic as S  # noqa: E402
from gbx_lm_amd.decode import FusedDecodeRunner  # noqa: E402
import bench  # noqa: E402

cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
caches = bench.make_caches(model, 1, int(os.environ.get("CAPACITY", "512")), dev)
L = _lib.lib()
L.gbx_debug_stamp_reset.restype = None
L.gbx_debug_stamp_reset()
runner = FusedDecodeRunner(model, caches, use_graph=True)
assert [len(d) for d in runner.da] == per
ctx = int(os.environ.get("CONTEXT", "128"))
for t in range(ctx):
    runner.step(torch.full((1,), t, device=dev, dtype=torch.long))
runner.start_greedy(torch.argmax(runner.logits, -1))
for _ in range(8):
    runner.step_greedy()
torch.cuda.synchronize()
stamps.zero_()
runner.step_greedy()
torch.cuda.synchronize()
runner.dattn_check()
s = stamps.cpu().numpy().reshape(SLOTS, 4096, 16)
END = {"dattn": 6, "dmlp": 7}
labels = {
    "dattn/producer": {0: "start", 1: "x, w, sums loads issued", 2: "barrier A (statistic)", 3: "barrier B (image; rows requested)", 4: "granule published", 12: "coop: q rotated (barrier)", 5: "coop: staged scores", 7: "coop: sweep done", 8: "coop: states in LDS", 13: "coop: chunk swept (barrier)", 15: "coop: merge weights (barrier)", 14: "coop: states published", 10: "H2 received", 11: "o image built", 6: "end"},
    "dattn/attention": {0: "start", 1: "K/V prefetch + o words issued", 5: "H1 received", 12: "RoPE done (barrier)", 13: "scores", 14: "values summed", 8: "merge input ready (coop: partial states received)", 9: "output published", 10: "H2 received", 6: "end"},
    "dmlp": {0: "start", 1: "sums, x, w loads issued", 9: "barrier A (statistic)", 10: "image written", 2: "barrier B", 3: "last gate|up rows requested", 4: "gate|up consumed, activations published", 5: "hand-off received (polling waves)", 6: "down image barrier", 7: "end"},
}
acc = {k: [] for k in labels}
spans = {"dattn": [], "dmlp": []}
gaps = {"dattn": [], "dmlp": []}
for li in ([int(v) for v in os.environ['LAYERS'].split(',')] if os.environ.get('LAYERS') else range(4, 29)):
    for ki, n in enumerate(("dattn", "dmlp")):
        r = s[base[li] + ki]
        live = r[:, 0] > 0
        t0 = r[live][:, 0].min()
        end = r[live][:, END[n]].max()
        spans[n].append((end - t0) / 100.0)
        nxt = s[base[li] + ki + 1]
        if (nxt[:, 0] > 0).any():
            gaps[n].append((nxt[nxt[:, 0] > 0][:, 0].min() - end) / 100.0)
        rel = (r.astype(np.float64) - t0) / 100.0
        rel[r == 0] = np.nan
        if n == "dattn":
            wg = np.arange(4096) // 16
            att = (wg >> 3) < 4
            acc["dattn/attention"].append(rel[live & att])
            acc["dattn/producer"].append(rel[live & ~att])
        else:
            acc[n].append(rel[live])
tot = 0.0
for n in ("dattn", "dmlp"):
    sp, gp = np.mean(spans[n]), np.mean(gaps[n])
    tot += sp + gp
    print(f"{n:6s} span {sp:6.2f} us   gap to next {gp:5.2f} us")
print(f"sum over a 2-bit layer: {tot:.2f} us")
for k, lab in labels.items():
    a = np.concatenate(acc[k], 0)
    print(f"-- {k} ({a.shape[0] // len(acc[k])} waves per launch)")
    for i, nm in lab.items():
        c = a[:, i]
        if np.all(np.isnan(c)):
            continue
        print(f"   s{i:<2d} {nm:46s} min {np.nanmin(c):6.2f}  med {np.nanmedian(c):6.2f}  p90 {np.nanpercentile(c, 90):6.2f}  max {np.nanmax(c):6.2f}")
li = int(os.environ.get("LAYERS", "10").split(",")[0])
r = s[base[li] + 1].astype(np.float64)
t0 = r[r[:, 0] > 0][:, 0].min()
rel = (r - t0) / 100.0
rel[r == 0] = np.nan
rel = rel.reshape(256, 16, 16)
cols = (0, 1, 9, 10, 2, 3, 4, 5, 6, 7)
print(f"dmlp, layer {li}: median over workgroups by wave index, stamps {cols}")
import warnings
warnings.simplefilter("ignore")
for w in range(16):
    c = rel[:, w]
    print(f"   wave {w:2d}: " + "  ".join(f"s{i}={np.nanmedian(c[:, i]):5.2f}" for i in cols) + f"   (end max {np.nanmax(c[:, 7]):5.2f})")
