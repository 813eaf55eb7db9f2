import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = "cuda:0"
b, hq, hkv, l, d = 8, 32, 8, 4096, 128
q = torch.randn((b, l, hq, d), device=dev).bfloat16().permute(0, 2, 1, 3)
k = torch.randn((b, hkv, l, d), device=dev).bfloat16()
v = torch.randn((b, hkv, l, d), device=dev).bfloat16()
for _ in range(3):
    ops.attn_prefill(q, k, v, 0, d ** -0.5)
torch.cuda.synchronize()
