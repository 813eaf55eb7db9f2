"""Prompt pass of a single short sequence (Llama-3-8B shapes, bpw 4.0): eager vs the graph-replayed bucket (decode.PromptRunner)."""
import sys, time, os, torch
sys.path.insert(0, '.')
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.models.cache import make_prompt_cache
from gbx_lm_amd.decode import prompt_step
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy(os.environ.get("STRAT", "S40"), cfg["num_hidden_layers"], 128), device=dev, seed=0)
from gbx_lm_amd import ops
for T in (16, 128, 256, 512, 1024):
    y = torch.randint(1, 1000, (T,), device=dev)
    for mode in ("eager", "graph", "eager+images", "graph+images"):
        if mode == "eager+images":
            print("  resident images:", ops.dense_cache_prewarm(model) >> 20, "MiB")
            model.__dict__.pop("_prompt_runners", None)
        if mode == "eager":
            ops.dense_cache_clear(); model.__dict__.pop("_prompt_runners", None)
        ts = []
        for rep in range(5):
            cache = make_prompt_cache(model)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                lg = model.last_logits(y[None], cache) if mode.startswith("eager") else prompt_step(model, y, cache)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        if lg is None:                      # prompt_step declines prompts beyond its largest bucket: the caller runs the eager pass
            print(f"T={T:5d} {mode}: no bucket (> 128 tokens): eager pass is used")
            continue
        print(f"T={T:5d} {mode}: prompt pass {min(ts[1:]):6.2f} ms (first call {ts[0]:.1f} ms)")
