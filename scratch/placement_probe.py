"""Does WHERE the decode runner's buffers land change the token rate?  (profiles/r06_dmlp_mx3.txt: with ~170 MB of 3-bit planes allocated in front of
them, every launch of the S22 token ran ~1.5 % longer.)  One model, decode legs behind dummy allocations of different sizes (kept alive)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gbx_lm_amd import synthetic as S
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
keep = []
for pad_mb in (0, 0, 1, 7, 33, 64, 170, 171, 250, 0):
    if pad_mb:
        keep.append(torch.empty((pad_mb << 20,), dtype=torch.uint8, device=dev))
    r = [bench.decode_leg(model, cfg, dev, 1, 128, 20, 5)["tok_s"] for _ in range(2)]
    print(f"pad +{pad_mb:4d} MB (total {sum(k.numel() for k in keep) >> 20:4d} MB): {r[0]:7.1f} {r[1]:7.1f} tok/s", flush=True)
    torch.cuda.empty_cache()
