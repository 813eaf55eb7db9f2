"""Does alternating between TWO instantiated copies of the token's graph remove the ~9 us between consecutive replays of ONE exec?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.decode import FusedDecodeRunner
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
caches = bench.make_caches(model, 1, 512, dev)
r = FusedDecodeRunner(model, caches)
for t in range(128):
    lg = r.step(torch.tensor([t], device=dev))
r.start_greedy(torch.argmax(lg, -1))
g1 = r._graph_for(True, False)[0]
del r._graphs[(True, False)]
g2 = r._graph_for(True, False)[0]
assert g1 is not g2
def run(graphs, n=40):
    for c in caches: c.offset = 128
    r.start_greedy(torch.tensor([5], device=dev))
    for i in range(5): graphs[i % len(graphs)].replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): graphs[i % len(graphs)].replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rep in range(3):
    print(f"one exec: {run([g1]):7.1f} us/token   two execs alternating: {run([g1, g2]):7.1f} us/token", flush=True)
