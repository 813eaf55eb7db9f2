"""QuantizedLinear GEMMs at prompt sizes between the decode batch and the long-prompt path (4-bit, 8B shapes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, _lib
dev = 'cuda:0'
def run(m, n, k, b=4, g=128, iters=20):
    qw = torch.randint(0, 2**32, (n, k // 32 * b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16); ze = (-sc * 1.5).to(torch.bfloat16)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16); y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=b, out=y)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=b, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    kid = _lib.lib().gbx_qlinear_kernel_id(m, n, k, b, g, 1)
    print(f"M={m:6d} N={n:6d} K={k:6d}: {ms * 1e3:9.1f} us  {2 * m * n * k / ms / 1e9:8.1f} TFLOP/s  kernel {kid}")
for m in (128, 256, 512, 1024, 2048, 3072, 4096, 8192):
    for (n, k) in ((4096, 4096), (14336, 4096), (4096, 14336)):
        run(m, n, k)
