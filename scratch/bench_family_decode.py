"""Greedy batch-1 decode of a registry family at public shapes (synthetic 'S22' mix) through generate_step:
python scratch/bench_family_decode.py QWEN3_8B | QWEN3_30B_A3B | LLAMA3_8B"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import synthetic as S, utils as U
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "QWEN3_8B"
cfg = getattr(S, name)
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
torch.cuda.synchronize()
prompt = torch.arange(64) % cfg["vocab_size"]
n, t0 = 0, None
for tok, lp, _ in U.generate_step(prompt, model, max_tokens=136):
    n += 1
    if n == 8:
        torch.cuda.synchronize(); t0 = time.perf_counter()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{name} S22: {(n - 8) / dt:.1f} tok/s ({dt / (n - 8) * 1e3:.2f} ms/token) through generate_step, runner {type(model._decode_runner).__name__}")
