import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = 'cuda:0'
m, n, k = int(os.environ.get("M", "32")), 128256, 4096
ws = [(torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
for i in range(3): ops.dense_linear(x, ws[i % 3])
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for i in range(12): y = ops.dense_linear(x, ws[i % 3])
gr.replay(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 12
print(f"M={m} dense head {n} x {k}: {us:.1f} us = {n * k * 2 / us / 1e6:.2f} TB/s")
