"""gbx_attn_prefill vs the torch SDPA route it displaces, cfg4 shape (B 8, T 4096, 32 / 8 heads of 128)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
from gbx_lm_amd.models.base import create_causal_mask
dev = "cuda:0"
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (b, hq, hkv, l, d) in [(8, 32, 8, 4096, 128), (1, 32, 8, 4096, 128), (1, 32, 8, 512, 128), (8, 32, 8, 1024, 128), (8, 16, 8, 4096, 64)]:
    q = torch.randn((b, l, hq, d), device=dev).bfloat16().permute(0, 2, 1, 3)
    k = torch.randn((b, hkv, l, d), device=dev).bfloat16()
    v = torch.randn((b, hkv, l, d), device=dev).bfloat16()
    mask = create_causal_mask(l, 0, device=dev).to(torch.bfloat16)
    sc = d ** -0.5
    flops = 4 * b * hq * l * l * d / 2
    def sdpa():
        return F.scaled_dot_product_attention(q, k.repeat_interleave(hq // hkv, 1), v.repeat_interleave(hq // hkv, 1), attn_mask=mask, scale=sc)
    def sdpa_c():
        return F.scaled_dot_product_attention(q, k.repeat_interleave(hq // hkv, 1), v.repeat_interleave(hq // hkv, 1), is_causal=True, scale=sc)
    def hip():
        return ops.attn_prefill(q, k, v, 0, sc)
    a, c, h = t(sdpa), t(sdpa_c), t(hip)
    print(f"B{b} Hq{hq} Hkv{hkv} L{l} D{d}: sdpa+mask {a:8.3f} ms ({flops/a/1e9:6.0f} TF)  sdpa causal {c:8.3f} ms ({flops/c/1e9:6.0f} TF)  hip {h:8.3f} ms ({flops/h/1e9:6.0f} TF)")
