"""Greedy batch-1 decode of Qwen3-30B-A3B shapes (128 experts, top-8; synthetic 'S22' mix) through generate_step:
graph-replayed family-agnostic runner + expert-indexed GEMV launches."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import synthetic as S, utils as U
dev = torch.device("cuda:0")
cfg = S.QWEN3_30B_A3B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
torch.cuda.synchronize()
print("weights GB:", sum(b.numel() * b.element_size() for b in model.buffers()) / 1e9)
prompt = torch.arange(64) % cfg["vocab_size"]
n = 0
t0 = None
for tok, lp, _ in U.generate_step(prompt, model, max_tokens=72):
    n += 1
    if n == 8:
        torch.cuda.synchronize(); t0 = time.perf_counter()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"qwen3-30b-a3b shapes S22: {(n - 8) / dt:.1f} tok/s ({dt / (n - 8) * 1e3:.2f} ms/token), runner {type(model._decode_runner).__name__}")
