"""where the per-token gap of generate_step comes from: step_gen replays with / without the per-token clone, D2H copy, event"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.decode import FusedDecodeRunner
dev = "cuda:0"
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
caches = bench.make_caches(model, 1, 1024, dev)
r = FusedDecodeRunner(model, caches)
y = torch.tensor([5], device=dev)
for t in range(128):
    y, lp, w = r.step_gen(y)
torch.cuda.synchronize()
pin = [torch.zeros((4,), dtype=torch.long, pin_memory=True) for _ in range(3)]
side = torch.cuda.Stream(device=dev)
def run(mode, n=200):
    global y
    for c in caches: c.offset = 128
    r._gen_next = None
    y = torch.tensor([5], device=dev)
    evs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prev = None
    for i in range(n):
        y, lp, w = r.step_gen(y)
        if "clone" in mode: keep = lp.view(-1).clone()
        if "copy" in mode:
            pin[i % 3].copy_(w, non_blocking=True)
        if "sidecopy" in mode:
            e = torch.cuda.Event(); e.record()
            with torch.cuda.stream(side):
                side.wait_event(e)
                pin[i % 3].copy_(w, non_blocking=True)
                if "clone" in mode: keep = lp.view(-1).clone()
        if "event" in mode:
            ev = torch.cuda.Event(); ev.record(side if "sidecopy" in mode else None)
            if prev is not None: prev.synchronize()
            prev = ev
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{mode:32s}: {(t1 - t0) / n * 1e6:7.1f} us/token = {n / (t1 - t0):7.1f} tok/s")
for m in ("replay only", "replay only", "clone", "copy", "clone copy", "clone copy event", "sidecopy event", "sidecopy(clone) event"):
    run(m.replace("sidecopy(clone)", "sidecopy clone") if "(" in m else m)
