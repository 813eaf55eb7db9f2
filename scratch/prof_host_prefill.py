"""Where the host time of a short prompt pass goes (cProfile over 5 passes of a 128-token prompt, Llama-3-8B S40)."""
import os, sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.models.cache import make_prompt_cache
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S40", cfg["num_hidden_layers"], 128), device=dev, seed=0)
T = int(os.environ.get("T", 128))
toks = (torch.arange(T, device=dev) % cfg["vocab_size"]).view(1, T)
def run():
    cache = make_prompt_cache(model)
    with torch.no_grad():
        model.model(toks, cache=cache)
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): run()
t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
print(f"host {1e3 * (t1 - t0) / 5:.2f} ms per pass, + {1e3 * (t2 - t1):.2f} ms drain")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): run()
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:4500])
