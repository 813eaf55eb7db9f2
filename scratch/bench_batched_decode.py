"""cfg5 on one GPU: synchronous batched greedy decode (B sequences, same position) of Llama-3-8B S22 through the
graph-replayed DecodeRunner (torch glue + split-K MFMA QuantizedLinear).  tokens/s = B / step time."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.decode import DecodeRunner
from gbx_lm_amd.models.cache import make_prompt_cache
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx, steps = 128, 32
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
toks = (torch.arange(B * ctx, device=dev) % cfg["vocab_size"]).view(B, ctx)
with torch.no_grad():
    cache = make_prompt_cache(model)
    logits = model(toks, cache=cache)[:, -1, :]
    runner = DecodeRunner(model, cache)
    y = torch.argmax(logits, -1)
    for _ in range(4): y = torch.argmax(runner.step(y), -1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): y = torch.argmax(runner.step(y), -1)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print(f"batched decode B={B} context={ctx}: {ms:.3f} ms/step = {B/ms*1e3:,.0f} tok/s")
