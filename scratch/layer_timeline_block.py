"""Timeline of one decoded token with the one-launch block (round 6: dblock), from in-kernel s_memrealtime stamps.
    scratch/build_stamps.sh
    GBX_QLINEAR_LIB=scratch/stamps/libgbx_stamps.so python scratch/layer_timeline_block.py"""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, ".")
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
NL = 32
REG = 4096 * 16
from gbx_lm_amd import synthetic as S0
bits = [S0.strategy_bits("S22", i, NL) for i in range(NL)]
per = [1 if b == 2 else 3 for b in bits]
SLOTS = sum(per)
base = np.cumsum([0] + per)
stamps = torch.zeros(SLOTS * REG, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
os.environ["GBX_STAMP_SLOTS"] = str(SLOTS)

from gbx_lm_amd import _lib, synthetic as S  # noqa: E402
from gbx_lm_amd.decode import FusedDecodeRunner  # noqa: E402
import bench  # noqa: E402

cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
caches = bench.make_caches(model, 1, int(os.environ.get("CAPACITY", "768")), dev)
L = _lib.lib()
L.gbx_debug_stamp_reset.restype = None
runner = FusedDecodeRunner(model, caches, use_graph=False)
assert [1 if d is not None else 3 for d in runner.db] == per
ctx = int(os.environ.get("CONTEXT", "128"))
for t in range(ctx):
    runner.step(torch.full((1,), t, device=dev, dtype=torch.long))
# the stamp region is chosen per launch at ENQUEUE time: capture the greedy chain now so that slot i = launch i of the token
L.gbx_debug_stamp_reset()
runner.use_graph = True
runner.start_greedy(torch.argmax(runner.logits, -1))
for _ in range(8):
    runner.step_greedy()
torch.cuda.synchronize()
stamps.zero_()
runner.step_greedy()
torch.cuda.synchronize()
runner.dattn_check()
s = stamps.cpu().numpy().reshape(-1, 4096, 16)
# the capture's warm-up passes and the capture itself each consumed SLOTS regions (round-robin): the replayed graph's launches are the LAST pass
labels = {
    "producer": {0: "start", 1: "x, w, sums loads issued", 2: "barrier B", 3: "granule published", 4: "H2 received (waves 0, 1)",
                 5: "o_proj sums issued", 6: "ARRIVE (sums acknowledged, W2 issued)", 7: "x' received", 8: "image + every piece landed (barrier)", 13: "(diag) B operands in registers", 14: "(diag) own tiles done", 15: "(diag) partials exchanged",
                 9: "gate|up consumed, activations published", 10: "hand-off received (polling waves)", 11: "down image barrier", 12: "end"},
    "attention": {0: "start", 1: "K/V prefetch + o words + W0 issued", 2: "H1 received (waves 0, 1)", 3: "output published", 4: "H2 received (waves 0, 1)",
                  5: "o_proj sums issued", 6: "ARRIVE", 13: "leader: 32 arrivals", 14: "leader: 8 flags", 15: "leader: x' published", 7: "x' received",
                  8: "image + every piece landed (barrier)", 9: "gate|up consumed, activations published", 10: "hand-off received (polling waves)",
                  11: "down image barrier", 12: "end"},
}
acc = {k: [] for k in labels}
spans, gaps = [], []
for li in range(4, 29):
    r = s[base[li]]
    live = r[:, 0] > 0
    t0 = r[live][:, 0].min()
    end = r[live][:, 12].max()
    spans.append((end - t0) / 100.0)
    nxt = s[base[li] + 1]
    if (nxt[:, 0] > 0).any():
        gaps.append((nxt[nxt[:, 0] > 0][:, 0].min() - end) / 100.0)
    rel = (r.astype(np.float64) - t0) / 100.0
    rel[r == 0] = np.nan
    wg = np.arange(4096) // 16
    att = (wg >> 3) < 4
    acc["attention"].append(rel[live & att])
    acc["producer"].append(rel[live & ~att])
print(f"dblock span {np.mean(spans):6.2f} us (min {np.min(spans):.2f}, max {np.max(spans):.2f})   gap to next {np.mean(gaps):5.2f} us   -> {np.mean(spans) + np.mean(gaps):.2f} us per 2-bit block")
for k, lab in labels.items():
    a = np.concatenate(acc[k], 0)
    print(f"-- {k} workgroups ({a.shape[0] // len(acc[k])} waves per launch)")
    for i, nm in lab.items():
        c = a[:, i]
        if np.all(np.isnan(c)):
            continue
        print(f"   s{i:<2d} {nm:46s} min {np.nanmin(c):6.2f}  med {np.nanmedian(c):6.2f}  p90 {np.nanpercentile(c, 90):6.2f}  max {np.nanmax(c):6.2f}")

# ---- who are the stragglers?  per member index r (median over the 8 XCDs and the layers) and per XCD slot: ARRIVE (s6) and end (s12)
rel_all = []
for li in range(4, 29):
    r_ = s[base[li]].astype(np.float64)
    t0 = r_[r_[:, 0] > 0][:, 0].min()
    rr = (r_ - t0) / 100.0
    rr[r_ == 0] = np.nan
    rel_all.append(rr.reshape(256, 16, 16))
rel_all = np.stack(rel_all)                       # [layers, workgroup, wave, stamp]
wg_arrive = np.nanmax(rel_all[:, :, :, 6], axis=2)  # last wave of the workgroup
wg_end = np.nanmax(rel_all[:, :, :, 12], axis=2)
wg_pub = np.nanmax(rel_all[:, :, :, 3], axis=2)
member = np.arange(256) >> 3
slot = np.arange(256) & 7
print("by member r: granule/output published (s3) | ARRIVE (s6) | end (s12), medians over XCDs and layers")
for m in range(32):
    sel = member == m
    print(f"   r={m:2d}: {np.nanmedian(wg_pub[:, sel]):5.2f} | {np.nanmedian(wg_arrive[:, sel]):5.2f} | {np.nanmedian(wg_end[:, sel]):5.2f}")
print("by blockIdx % 8: ARRIVE | end")
for q in range(8):
    sel = slot == q
    print(f"   slot {q}: {np.nanmedian(wg_arrive[:, sel]):5.2f} | {np.nanmedian(wg_end[:, sel]):5.2f}")
last_arr = np.nanargmax(wg_arrive, axis=1)
last_end = np.nanargmax(wg_end, axis=1)
print("the LAST workgroup to arrive, per layer (block index, member):", [(int(b), int(b) >> 3) for b in last_arr])
print("the LAST workgroup to end, per layer (block index, member):", [(int(b), int(b) >> 3) for b in last_end])
