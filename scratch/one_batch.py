import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = 'cuda:0'
m, n, k, b, g = 32, int(sys.argv[1]), int(sys.argv[2]), 2, 128
qw = torch.randint(0, 2**32, (n, k // 32 * b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16); ze = (-sc * 1.5).to(torch.bfloat16)
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
for it in range(5):
    ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=b)
torch.cuda.synchronize()
