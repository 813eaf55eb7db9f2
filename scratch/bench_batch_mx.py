"""M = 32, 2-bit: the product batch GEMM (gbx_qlinear_forward) against the experimental FP4 x FP8 form (gbx_exp_batch_mx_forward, image prebuilt) on the
8B shapes; rotating weight copies (> 600 MB) so that no launch reads a cached matrix; graph of 50 launches."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, _lib
dev = 'cuda:0'
L = _lib.exp_lib()


def run(m, n, k, iters=50):
    b, g = 2, 128
    copies = max(2, int(600e6 // (n * k * b // 8)) + 1)
    ws = []
    for _ in range(copies):
        qw = torch.randint(0, 2**32, (n, k // 32 * b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
        sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16); ze = (-sc * 1.5).to(torch.bfloat16)
        ws.append((qw, sc, ze))
    x = torch.randn(m, k, device=dev).to(torch.bfloat16); y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    img = torch.empty(L.gbx_exp_batch_mx_image_bytes(k), dtype=torch.uint8, device=dev)
    aux = torch.empty(L.gbx_exp_batch_mx_aux_bytes(k), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check_exp(L.gbx_exp_batch_mx_image(x.data_ptr(), m, k, img.data_ptr(), aux.data_ptr(), st))

    class _M:                               # the product's k-split partials path takes a module
        pass

    def prod(i):
        ops.quantized_matmul(x, *ws[i % copies], group_size=g, bits=b, out=y)

    ks = L.gbx_exp_batch_mx_slices(n, k)
    sl = torch.empty(ks, m, n, dtype=torch.float32, device=dev) if ks > 1 else None

    def mx(i):
        qw, sc, ze = ws[i % copies]
        _lib.check_exp(L.gbx_exp_batch_mx_forward(img.data_ptr(), aux.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), y.data_ptr(),
                                                  sl.data_ptr() if sl is not None else None, m, n, k, torch.cuda.current_stream().cuda_stream))
    out = []
    for fn in (prod, mx):
        for i in range(3): fn(i)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(iters): fn(i)
        gr.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / iters)
    wb = n * k * b / 8 + 2 * n * (k // g) * 2
    print(f"M={m:3d} N={n:6d} K={k:6d}: product {out[0]:6.1f} us ({wb / out[0] / 1e3:6.1f} GB/s)   FP4 x FP8 {out[1]:6.1f} us ({wb / out[1] / 1e3:6.1f} GB/s, k ranges {ks})", flush=True)


for (n, k) in [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336)]:
    run(32, n, k)
