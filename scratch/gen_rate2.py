"""per-token wall-clock deltas of generate_step (diagnostic)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.utils import generate_step
dev = "cuda:0"
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
prompt = (torch.arange(128) * 7 + 3) % 32000
for rep in range(3):
    ts = []
    for i, (tok, logprobs, _) in enumerate(generate_step(prompt, model, max_tokens=300)):
        ts.append(time.perf_counter())
    d = [(b - a) * 1e6 for a, b in zip(ts, ts[1:])]
    s = sorted(d[10:])
    print(f"run {rep}: first 12 deltas us: {[int(x) for x in d[:12]]}")
    print(f"   tokens 10..: median {s[len(s)//2]:.0f} p10 {s[len(s)//10]:.0f} p90 {s[len(s)*9//10]:.0f} max {s[-1]:.0f} mean {sum(s)/len(s):.0f}; spikes > 3000 us at {[i+10 for i, x in enumerate(d[10:]) if x > 3000]}")
import cProfile, pstats
pr = cProfile.Profile()
g = generate_step(prompt, model, max_tokens=300)
for _ in range(20): next(g)
pr.enable()
for _ in range(250): next(g)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
