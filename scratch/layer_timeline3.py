"""Timeline of one decoded token with the three-launch block (round 5: dattn, gate|up, down), from in-kernel s_memrealtime stamps.

    scratch/build_stamps.sh
    GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so python scratch/layer_timeline3.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
dev = torch.device("cuda:0")
PER_LAYER, NL = 3, 32
SLOTS = PER_LAYER * NL
REG = 4096 * 16
stamps = torch.zeros(SLOTS * REG, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
os.environ["GBX_STAMP_SLOTS"] = str(SLOTS)
os.environ["GBX_DATTN"] = "1"

from gbx_lm_amd import _lib, synthetic as S  # noqa: E402
from gbx_lm_amd.decode import FusedDecodeRunner  # noqa: E402
import bench  # noqa: E402

cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
caches = bench.make_caches(model, 1, 512, dev)
L = _lib.lib()
L.gbx_debug_stamp_reset.restype = None
L.gbx_debug_stamp_reset()
runner = FusedDecodeRunner(model, caches, use_graph=True)
assert all(d is not None for d in runner.da)
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
bits = [blk.mlp.gate_proj.bits for blk in model.model.layers]
names = ["dattn", "gateup", "down"]
labels = {
    "dattn/producer": {0: "start", 1: "x, w, ssq loads issued", 2: "statistic", 3: "image + barrier (rows, o words requested)", 4: "granule published", 10: "H2 received + barrier", 11: "o image built", 6: "end"},
    "dattn/attention": {0: "start", 1: "K/V prefetch + o words issued", 5: "H1 received + barrier", 12: "RoPE done (barrier)", 15: "(stamps build) all prefetched K / V landed", 13: "scores (K landed)", 14: "values summed (V landed)", 8: "sweep + in-wave merge done", 9: "output published", 10: "H2 received + barrier", 11: "o image built", 6: "end"},
    "gateup": {0: "start", 1: "x, w loads issued", 8: "acc combined", 9: "statistic", 10: "image written", 2: "barrier", 4: "first batch of the last pair consumed", 6: "end"},
    "down": {0: "start", 1: "x + batch 0 issued", 2: "barrier", 3: "lane x ready", 4: "batch 0 consumed", 6: "end"},
}
acc = {k: [] for k in labels}
spans = {n: [] for n in names}
gaps = {n: [] for n in names}
for li in range(NL):
    if bits[li] != 2 or not (3 <= li <= 28):
        continue
    for ki, n in enumerate(names):
        r = s[li * PER_LAYER + ki]
        live = r[:, 0] > 0
        t0 = r[live][:, 0].min()
        end = r[live][:, 6].max()
        spans[n].append((end - t0) / 100.0)
        nxt = s[li * PER_LAYER + ki + 1] if li * PER_LAYER + ki + 1 < SLOTS else None
        if nxt is not None and (nxt[:, 0] > 0).any():
            gaps[n].append((nxt[nxt[:, 0] > 0][:, 0].min() - end) / 100.0)
        rel = (r.astype(np.float64) - t0) / 100.0
        rel[r == 0] = np.nan
        if n == "dattn":
            wg = np.arange(4096) // 16
            role_att = (wg >> 3) < 4
            acc["dattn/attention"].append(rel[live & role_att])
            acc["dattn/producer"].append(rel[live & ~role_att])
        else:
            acc[n].append(rel[live])
tot = 0.0
for n in names:
    sp, gp = np.mean(spans[n]), np.mean(gaps[n]) if gaps[n] else float("nan")
    tot += sp + (gp if gp == gp else 0)
    print(f"{n:7s} span {sp:6.2f} us   gap to next {gp:5.2f} us")
print(f"sum over a 2-bit layer: {tot:.2f} us")
for k, lab in labels.items():
    a = np.concatenate(acc[k], 0)
    print(f"-- {k} ({a.shape[0] // len(acc[k])} waves per launch)")
    for i, nm in lab.items():
        c = a[:, i]
        if np.all(np.isnan(c)):
            continue
        print(f"   s{i:<2d} {nm:44s} min {np.nanmin(c):6.2f}  med {np.nanmedian(c):6.2f}  p90 {np.nanpercentile(c, 90):6.2f}  max {np.nanmax(c):6.2f}")

# ---- where does the producers' tail come from?  per wave index and per workgroup (one 2-bit layer, the 10th)
li = 10
r = s[li * PER_LAYER].astype(np.float64)
t0 = r[r[:, 0] > 0][:, 0].min()
rel = (r - t0) / 100.0
rel[r == 0] = np.nan
rel = rel.reshape(256, 16, 16)
prod = (np.arange(256) >> 3) >= 4
print("dattn producers, layer 10: median over workgroups, by wave index")
for w in range(16):
    c = rel[prod][:, w]
    print(f"   wave {w:2d}: start {np.nanmedian(c[:, 0]):5.2f}  loads issued {np.nanmedian(c[:, 1]):5.2f}  statistic {np.nanmedian(c[:, 2]):5.2f}  barrier {np.nanmedian(c[:, 3]):5.2f}  published {np.nanmedian(c[:, 4]):5.2f}  (max {np.nanmax(c[:, 4]):5.2f})")
wg_bar = np.nanmax(rel[:, :, 3], axis=1)
wg_start = np.nanmin(rel[:, :, 0], axis=1)
wg_issue = np.nanmax(rel[:, :, 1], axis=1)
order = np.argsort(-np.where(prod, wg_bar, -1))[:12]
print("slowest producer workgroups (barrier passed):")
for b in order:
    print(f"   block {b:3d} (group slot {b % 8}, member {b >> 3}): first wave start {wg_start[b]:5.2f}, last wave's loads issued {wg_issue[b]:5.2f}, barrier {wg_bar[b]:5.2f}, last publish {np.nanmax(rel[b, :, 4]):5.2f}")
print("by member index r (median over the 8 groups): barrier passed")
print("   " + "  ".join(f"{m}:{np.nanmedian(wg_bar[np.arange(256) >> 3 == m]):.2f}" for m in range(32)))

for ki, (nm, cols) in enumerate([("dattn", None), ("gateup", (0, 1, 8, 9, 10, 11, 2, 4, 6)), ("down", (0, 1, 2, 3, 4, 6))]):
    if cols is None:
        continue
    r = s[li * PER_LAYER + ki].astype(np.float64)
    t0 = r[r[:, 0] > 0][:, 0].min()
    rel = (r - t0) / 100.0
    rel[r == 0] = np.nan
    rel = rel.reshape(256, 16, 16)
    print(f"{nm}, layer {li}: median over workgroups by wave index, stamps {cols}")
    for w in range(16):
        c = rel[:, w]
        print(f"   wave {w:2d}: " + "  ".join(f"s{i}={np.nanmedian(c[:, i]):5.2f}" for i in cols) + f"   (end max {np.nanmax(c[:, 6]):5.2f})")
