"""M=32 (and 8) 2-bit skinny GEMM timing on the four 8B shapes; GBX_QLINEAR_LIB selects an experiment build."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = 'cuda:0'
def run(m, n, k, b, g=int(os.environ.get('GROUP', '128')), iters=50):
    copies = max(2, int(600e6 // (n * k * b // 8)) + 1)
    ws = []
    for _ in range(copies):
        qw = torch.randint(0, 2**32, (n, k // 32 * b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
        sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16); ze = (-sc * 1.5).to(torch.bfloat16)
        ws.append((qw, sc, ze))
    x = torch.randn(m, k, device=dev).to(torch.bfloat16); y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for i in range(3): ops.quantized_matmul(x, *ws[i % copies], group_size=g, bits=b, out=y)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(iters): ops.quantized_matmul(x, *ws[i % copies], group_size=g, bits=b, out=y)
    gr.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    wb = n * k * b / 8 + 2 * n * (k // g) * 2
    print(f"M={m:3d} N={n:6d} K={k:6d} b={b}: {us:7.1f} us  weights {wb / us / 1e3:7.1f} GB/s", flush=True)
bits = [int(v) for v in os.environ.get("BITS", "2").split(",")]
ms = [int(v) for v in os.environ.get("MS", "32").split(",")]
print(os.environ.get("GBX_QLINEAR_LIB", "default"))
for b in bits:
    for m in ms:
        for (n, k) in [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336)]:
            run(m, n, k, b)
