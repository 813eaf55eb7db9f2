"""batch GEMM (M = 32) against fp32 x . dequantised weights on the 8B shapes; GBX_QLINEAR_LIB selects an experiment build."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = 'cuda:0'
torch.manual_seed(0)
bits = int(os.environ.get("BITS", "2"))
for m in (32, 7):
    for (n, k) in [(28672, 4096), (4096, 14336), (16384, 4096), (6144, 4096), (24576, 1024)]:
        qw = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
        sc = (torch.rand((n, k // 128), device=dev) * 0.01 + 0.001).to(torch.bfloat16); ze = (-sc.float() * (2 ** bits - 1) / 2 * torch.rand_like(sc.float())).to(torch.bfloat16)
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        y = ops.quantized_matmul(x, qw, sc, ze, group_size=128, bits=bits).float()
        w = ops.dequantize(qw, sc, ze, 128, bits, dtype=torch.float32)
        ref = x.float() @ w.t()
        err = (y - ref).abs().max().item(); scale = ref.abs().mean().item()
        print(f"M={m} N={n} K={k} bits={bits}: max err {err:.4g} (mean |ref| {scale:.4g}) {'OK' if err < 0.02 * scale + 0.02 * ref.abs().max().item() * 0.1 else 'BAD'}")
