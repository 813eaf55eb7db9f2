"""Timeline of one decoded token's launches from in-kernel s_memrealtime stamps (100 MHz), graph replay, 8B S22.

    scratch/build_stamps.sh
    GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so python scratch/layer_timeline.py

Every stamped launch of a forward pass (five per layer in the 5-launch block) owns one region of the stamp buffer
(GBX_STAMP_SLOTS = launches per forward, so the regions line up across warm-up, capture and replay).  Prints, per kind of
launch averaged over the 2-bit layers: kernel span (first wave start -> last wave end), the phase medians relative to the
first start, and the gap to the next launch (last end -> next first start)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
dev = torch.device("cuda:0")
PER_LAYER = int(os.environ.get("PER_LAYER", "5"))
NL = 32
SLOTS = PER_LAYER * NL
REG = 4096 * 16
stamps = torch.zeros(SLOTS * REG, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
os.environ["GBX_STAMP_SLOTS"] = str(SLOTS)

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
s = stamps.cpu().numpy().reshape(SLOTS, -1, 16)

names5 = ["qkv", "attn", "o", "gateup", "down"]
names = names5 if PER_LAYER == 5 else [f"k{i}" for i in range(PER_LAYER)]
bits = [blk.mlp.gate_proj.bits for blk in model.model.layers]
rows = {n: [] for n in names}
gaps = {n: [] for n in names}
t_first = None
for li in range(NL):
    for ki, n in enumerate(names):
        r = s[li * PER_LAYER + ki]
        r = r[r[:, 0] > 0]
        if len(r) == 0:
            continue
        t0 = r[:, 0].min()
        if t_first is None:
            t_first = t0
        end = r[:, 6].max()
        nxt = li * PER_LAYER + ki + 1
        gap = None
        if nxt < SLOTS:
            rn = s[nxt]
            rn = rn[rn[:, 0] > 0]
            if len(rn):
                gap = (rn[:, 0].min() - end) / 100.0
        if bits[li] == 2 and 3 <= li <= 28:
            rel = (r - t0) / 100.0
            med = [np.median(rel[:, i][r[:, i] > 0]) if (r[:, i] > 0).any() else float("nan") for i in range(16)]
            mx = [rel[:, i][r[:, i] > 0].max() if (r[:, i] > 0).any() else float("nan") for i in range(16)]
            rows[n].append(((end - t0) / 100.0, med, mx, len(r), rel[:, 0].max()))
            if gap is not None:
                gaps[n].append(gap)
last = s[SLOTS - 1]
last = last[last[:, 0] > 0]
print(f"layers total: {(last[:, 6].max() - t_first) / 100.0:.1f} us for {NL} layers (first qkv start -> last down end)")
tot = 0.0
for n in names:
    if not rows[n]:
        continue
    span = np.mean([r[0] for r in rows[n]])
    med = np.nanmean(np.array([r[1] for r in rows[n]]), axis=0)
    mx = np.nanmean(np.array([r[2] for r in rows[n]]), axis=0)
    gp = np.mean(gaps[n]) if gaps[n] else float("nan")
    tot += span + (gp if gp == gp else 0)
    print(f"{n:7s} waves {rows[n][0][3]:5d}  span {span:6.2f} us  start-spread {np.mean([r[4] for r in rows[n]]):5.2f}  gap-to-next {gp:5.2f}")
    print("         stamp medians: " + "  ".join(f"s{i}={med[i]:.2f}" for i in (0, 1, 8, 9, 10, 11, 12, 2, 3, 4, 5, 6, 7)))
    print("         stamp maxima:  " + "  ".join(f"s{i}={mx[i]:.2f}" for i in range(7)))
print(f"sum over a 2-bit layer (spans + gaps): {tot:.2f} us")
print("gemv stamps: 0 start, 1 x + first batch issued, 2 prologue done (barrier), 3 ring issued, 4 loop done, 5 first drain consumed, 6 end")
print("attn stamps: 0 start, 1 K/V prefetch issued + pos loaded, 2 RoPE barrier, 3 sweep done, 4 merge barrier, 6 end")

# ---- placement: how many workgroups share a CU, and does the tail follow it? (stamp 7 = XCC id << 32 | HW_ID)
def placement(slot, nw, label):
    r = s[slot]
    r = r[r[:, 0] > 0]
    if len(r) % nw or not (r[:, 7] > 0).any():
        return
    t0 = r[:, 0].min()
    hw = r[:, 7]
    xcc = (hw >> 32) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    wg_cu = cuid.reshape(-1, nw)[:, 0]
    wg_end = ((r[:, 6] - t0) / 100.0).reshape(-1, nw).max(1)
    wg_pro = ((r[:, 2] - t0) / 100.0).reshape(-1, nw).max(1)
    ids, cnt = np.unique(wg_cu, return_counts=True)
    print(f"{label}: {len(wg_cu)} workgroups on {len(ids)} distinct CUs; workgroups per CU histogram {np.bincount(cnt)[1:]}; XCC histogram {np.bincount(xcc.reshape(-1, nw)[:, 0].astype(int))}")
    per = dict(zip(ids, cnt))
    share = np.array([per[c] for c in wg_cu])
    for k in sorted(set(share)):
        m = share == k
        print(f"   workgroups sharing their CU with {k - 1} other(s): n={m.sum():4d}  prologue done {wg_pro[m].mean():5.2f}  end mean {wg_end[m].mean():5.2f} max {wg_end[m].max():5.2f}")
    for x in range(8):
        m = xcc.reshape(-1, nw)[:, 0] == x
        if m.any():
            print(f"   XCC {x}: n={m.sum():3d} end mean {wg_end[m].mean():5.2f} max {wg_end[m].max():5.2f}")

if PER_LAYER == 5:
    li = 10
    placement(li * 5 + 0, 4, "qkv")
    placement(li * 5 + 2, 4, "o")
    placement(li * 5 + 3, 16, "gateup")
    placement(li * 5 + 4, 16, "down")

# ---- per-role view (gemv_ov_kernel: waves 0..3 of a workgroup are the prologue waves)
def roles(slot, nw, label, pw=4):
    r = s[slot]
    r = r[r[:, 0] > 0]
    if len(r) % nw:
        return
    t0 = r[:, 0].min()
    rel = ((r - t0) / 100.0).reshape(-1, nw, 16)
    for nm, sl in (("prologue waves", slice(0, pw)), ("loader waves", slice(pw, nw))):
        x = rel[:, sl, :].reshape(-1, 16)
        print(f"{label} {nm:15s}: " + "  ".join(f"s{i}={np.median(x[:, i]):.2f}/{x[:, i].max():.2f}" for i in range(7)) + "   (median/max)")

if PER_LAYER == 5 and os.environ.get("ROLES"):
    li = 10
    nws = [int(v) for v in os.environ["ROLES"].split(",")]
    roles(li * 5 + 0, nws[0], "qkv")
    roles(li * 5 + 2, nws[1], "o")
    roles(li * 5 + 3, nws[2], "gateup")
    roles(li * 5 + 4, nws[3], "down")

# ---- round 4: who ends last in the gate|up launch?  per wave-index (0..15) medians of the end stamp and of "batch 0 landed"
def by_wave(slot, nw, label):
    r = s[slot]
    r = r[r[:, 0] > 0]
    if len(r) % nw:
        return
    t0 = r[:, 0].min()
    rel = ((r - t0) / 100.0).reshape(-1, nw, 16)
    print(f"{label}: per wave index, median / max over workgroups of [start, batch 0 landed, end]")
    for w in range(nw):
        x = rel[:, w, :]
        print(f"   wave {w:2d}: start {np.median(x[:, 0]):5.2f}/{x[:, 0].max():5.2f}  landed {np.median(x[:, 11]):5.2f}/{x[:, 11].max():5.2f}  end {np.median(x[:, 6]):5.2f}/{x[:, 6].max():5.2f}")
    wg_end = rel[:, :, 6].max(1)
    print(f"   workgroup end: min {wg_end.min():.2f} p10 {np.percentile(wg_end, 10):.2f} median {np.median(wg_end):.2f} p90 {np.percentile(wg_end, 90):.2f} max {wg_end.max():.2f}")

if PER_LAYER == 5 and os.environ.get("BYWAVE"):
    by_wave(10 * 5 + 3, 16, "gateup")
    by_wave(10 * 5 + 4, 16, "down")
    by_wave(10 * 5 + 0, 16, "qkv")
