"""cfg4 prompt pass (B 8 x T 4096, Llama-3-8B S40) once warm, once measured: the program to put under rocprofv3."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import synthetic as S
from gbx_lm_amd.models.cache import make_prompt_cache
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
B, T = int(os.environ.get('B', 8)), int(os.environ.get('T', 4096))
model = S.build_model(cfg, S.make_strategy("S40", cfg["num_hidden_layers"], 128), device=dev, seed=0)
toks = (torch.arange(B * T, device=dev) % cfg["vocab_size"]).view(B, T)
def run():
    cache = make_prompt_cache(model)
    with torch.no_grad():
        for s0 in range(0, T, 2048):
            model.model(toks[:, s0:s0 + 2048], cache=cache)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    torch.cuda.synchronize(); t0 = time.time(); run(); torch.cuda.synchronize()
    print(f"pass {i}: {(time.time() - t0) * 1e3:.1f} ms")
