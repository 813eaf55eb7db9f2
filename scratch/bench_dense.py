"""cfg4 GEMMs: fused dequant-GEMM tiles vs dequantise-once + dense tiles (gbx_qlinear_forward vs gbx_qlinear_forward_ws)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import _lib
L = _lib.lib(); dev = 'cuda:0'
def run(m, n, k, b, g=128, iters=10):
    qw = torch.randint(0, 2**32, (n, k // 32 * b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16); ze = (-sc * 1.5).to(torch.bfloat16)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16); y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    need = L.gbx_qlinear_workspace_bytes(m, n, k, b, g); ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def fused(): _lib.check(L.gbx_qlinear_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, y.data_ptr(), m, n, k, b, g, 1, st))
    def dense(): _lib.check(L.gbx_qlinear_forward_ws(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, y.data_ptr(), m, n, k, b, g, 1, ws.data_ptr(), need, st))
    out = []
    for f in (fused, dense):
        for _ in range(2): f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out.append((ms, 2 * m * n * k / ms / 1e9))
    print(f"M={m:6d} N={n:6d} K={k:6d} b={b}: fused {out[0][0]*1e3:8.1f} us {out[0][1]:7.1f} TF | dense {out[1][0]*1e3:8.1f} us {out[1][1]:7.1f} TF")
for b in (4, 2):
    for (m, n, k) in ((16384, 4096, 4096), (16384, 1024, 4096), (16384, 14336, 4096), (16384, 4096, 14336), (4096, 4096, 4096), (32768, 4096, 4096)):
        run(m, n, k, b)
