"""A few decoded tokens of 8B S22 through FusedDecodeRunner WITHOUT a graph (eager launches): the workload of the PMC passes
(rocprofv3 --pmc serialises every dispatch; a replayed graph of 70 launches per token takes minutes there)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.decode import FusedDecodeRunner
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device="cuda:0", seed=0)
caches = bench.make_caches(model, 1, 256, "cuda:0")
r = FusedDecodeRunner(model, caches, use_graph=False)
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    r.step(torch.tensor([t + 5], device="cuda:0"))
torch.cuda.synchronize()
r.dattn_check()
print("ok")
