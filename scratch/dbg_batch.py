import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = 'cuda:0'
torch.manual_seed(0)
bits, m, n, k = 2, 32, 24576, 1024
qw = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
sc = (torch.rand((n, k // 128), device=dev) * 0.01 + 0.001).to(torch.bfloat16); ze = (-sc.float() * 1.5).to(torch.bfloat16)
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
y = ops.quantized_matmul(x, qw, sc, ze, group_size=128, bits=bits).float()
ref = x.float() @ ops.dequantize(qw, sc, ze, 128, bits, dtype=torch.float32).t()
bad = ((y - ref).abs() > 0.05) | ~torch.isfinite(y)
print("bad frac", bad.float().mean().item())
print("bad by m:", bad.float().mean(1).cpu().numpy().round(2))
bn = bad.float().mean(0).view(-1, 128).mean(0).cpu().numpy().round(2)
print("bad by n%128:", bn)
print("bad by workgroup (first 16):", bad.float().mean(0).view(-1, 128).mean(1)[:16].cpu().numpy().round(2))
print(y[0, :8], ref[0, :8])
