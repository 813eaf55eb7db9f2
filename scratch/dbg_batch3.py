import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = 'cuda:0'
torch.manual_seed(0)
bits, m, n, k = 2, 32, 24576, 1024
qw = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
sc = torch.ones((n, k // 128), device=dev, dtype=torch.bfloat16); ze = torch.zeros_like(sc)
w = ops.dequantize(qw, sc, ze, 128, bits, dtype=torch.float32)
badpos = []
for base in range(0, k, 32):
    x = torch.zeros(m, k, device=dev, dtype=torch.bfloat16)
    for i in range(m): x[i, base + i] = 1
    y = ops.quantized_matmul(x, qw, sc, ze, group_size=128, bits=bits).float()
    ref = x.float() @ w.t()
    bad = ((y - ref).abs() > 0.01).float().mean(1)
    for i in range(m):
        if bad[i] > 0: badpos.append((base + i, round(bad[i].item(), 2)))
print("bad k positions:", badpos[:64], "count", len(badpos))
# two rows hot at once: interaction
x = torch.zeros(m, k, device=dev, dtype=torch.bfloat16); x[0, 3] = 1; x[0, 700] = 1; x[5, 3] = 2
y = ops.quantized_matmul(x, qw, sc, ze, group_size=128, bits=bits).float(); ref = x.float() @ w.t()
print("two-hot err", (y - ref).abs().max().item())
x = torch.ones(m, k, device=dev, dtype=torch.bfloat16)
for rep in range(3):
    y = ops.quantized_matmul(x, qw, sc, ze, group_size=128, bits=bits).float(); ref = x.float() @ w.t()
    print("ones: y/ref", (y[0, :6] / ref[0, :6]).cpu().numpy(), "rows equal", bool((y == y[0:1]).all()))
x = torch.ones(m, k, device=dev, dtype=torch.bfloat16) * 0; x[:, :128] = 1
y = ops.quantized_matmul(x, qw, sc, ze, group_size=128, bits=bits).float(); ref = x.float() @ w.t()
print("first chunk ones: y", y[0, :6].cpu().numpy(), "ref", ref[0, :6].cpu().numpy())
