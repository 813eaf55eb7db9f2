"""tokens/s of the PRODUCT API: generate_step (utils.py:217-338 of the reference) on synthetic Llama-3-8B S22 weights, host in the loop --
one token id (+ two status words) crosses PCIe per token, the host enqueues step n + 1 before it reads token n.  Compare bench.py's device rate."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.utils import generate_step
dev = "cuda:0"
strategy = sys.argv[1] if len(sys.argv) > 1 else "S22"
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy(strategy, cfg["num_hidden_layers"], 128), device=dev, seed=0)
prompt = (torch.arange(128) * 7 + 3) % 32000
for rep in range(3):
    gen = generate_step(prompt, model, max_tokens=ntok)
    toks = []
    t_first = None
    for i, (tok, logprobs, _) in enumerate(gen):
        if i == 4:
            torch.cuda.synchronize(); t_first = time.perf_counter()      # (the first tokens carry the prompt pass and the graph capture)
        toks.append(tok)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    n = len(toks) - 5
    print(f"{strategy} generate_step: {n / (t1 - t_first):8.1f} tok/s over {n} tokens ({(t1 - t_first) / n * 1e6:.1f} us/token), run {rep}")
