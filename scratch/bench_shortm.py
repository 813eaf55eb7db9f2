"""QuantizedLinear at short-prompt row counts: the fused dequant tiles (what runs today below 1024 rows) vs the dense GEMM on a resident
bf16 image (gbx_dense_forward_ws, k split over workgroups)."""
import sys, torch
sys.path.insert(0, '.')
from gbx_lm_amd import ops, _lib
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dev = 'cuda:0'
L = _lib.lib()
def mod(n, k, bits=4, g=128):
    m = QuantizedLinear(k, n, False, g, bits)
    m._buffers["qweight"] = torch.randint(0, 2**32, (n, k // 32 * bits), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16)
    m._buffers["scales"] = sc; m._buffers["zeros"] = (-sc * 7.5).to(torch.bfloat16)
    return m
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
shapes = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]
for M in (64, 128, 256, 512, 768):
    tot_a = tot_b = 0.0
    for n, k in shapes:
        m = mod(n, k)
        x = torch.randn(M, k, device=dev).to(torch.bfloat16)
        ta = timeit(lambda: m(x))
        dense = ops.dequantize(m.qweight, m.scales, m.zeros, 128, 4)
        out = torch.empty((M, n), dtype=torch.bfloat16, device=dev)
        part = L.gbx_dense_workspace_bytes(M, n, k)
        ws = torch.empty((max(part, 16),), dtype=torch.uint8, device=dev)
        def fb():
            rc = L.gbx_dense_forward_ws(x.data_ptr(), dense.data_ptr(), None, out.data_ptr(), M, n, k, 1, ws.data_ptr() if part else None, part, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, L.gbx_last_error()
        tb = timeit(fb)
        fl = 2.0 * M * n * k
        print(f"M={M:4d} N={n:6d} K={k:6d}: fused tiles {ta:7.1f} us ({fl/ta/1e6:6.0f} TF/s)   dense image {tb:7.1f} us ({fl/tb/1e6:6.0f} TF/s)")
        tot_a += ta; tot_b += tb
    print(f"M={M:4d} per layer (q|k|v as one, o, gate|up as one, down): fused {tot_a:.0f} us  dense {tot_b:.0f} us  -> x32 = {tot_a*32/1e3:.1f} / {tot_b*32/1e3:.1f} ms")
