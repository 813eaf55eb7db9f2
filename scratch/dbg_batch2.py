import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = 'cuda:0'
torch.manual_seed(0)
bits, m, n, k = 2, 32, 24576, 1024
qw = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
sc = torch.ones((n, k // 128), device=dev, dtype=torch.bfloat16); ze = torch.zeros_like(sc)
w = ops.dequantize(qw, sc, ze, 128, bits, dtype=torch.float32)
for mode in ("ones", "onehot"):
    x = torch.zeros(m, k, device=dev, dtype=torch.bfloat16)
    if mode == "ones": x[:] = 1
    else:
        for i in range(m): x[i, i * 32 + 5] = 1
    y = ops.quantized_matmul(x, qw, sc, ze, group_size=128, bits=bits).float()
    ref = x.float() @ w.t()
    print(mode, "bad frac", ((y - ref).abs() > 0.01).float().mean().item())
    print(" y  ", y[:4, :10].cpu().numpy())
    print(" ref", ref[:4, :10].cpu().numpy())
# scales test: x ones, scales = n-dependent
sc2 = (torch.arange(n, device=dev).float()[:, None] % 7 + 1).expand(n, k // 128).contiguous().to(torch.bfloat16)
x = torch.ones(m, k, device=dev, dtype=torch.bfloat16)
y = ops.quantized_matmul(x, qw, sc2, ze, group_size=128, bits=bits).float()
ref = x.float() @ ops.dequantize(qw, sc2, ze, 128, bits, dtype=torch.float32).t()
print("scales: y/ref", (y[0, :16] / ref[0, :16]).cpu().numpy())
