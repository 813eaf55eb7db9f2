"""In-kernel phase stamps of mfma_batch.hip (experiment build with -DBX_STAMPS): s_memtime per phase."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda:0'
stamps = torch.zeros(2048 * 12 * 24, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
from gbx_lm_amd import ops
m, n, k, b, g = 32, int(sys.argv[1]), int(sys.argv[2]), 2, 128
qw = torch.randint(0, 2**32, (n, k // 32 * b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16); ze = (-sc * 1.5).to(torch.bfloat16)
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
big = torch.empty(300 * 2**20, dtype=torch.uint8, device=dev)
for it in range(3):
    big.zero_()                    # flush caches
    stamps.zero_(); torch.cuda.synchronize()
    ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=b); torch.cuda.synchronize()
allw = stamps.cpu().numpy().reshape(-1, 12, 24)
allw = allw[allw[:, 0, 0] > 0]
t0 = allw[:, :, 0].min(axis=1)[:, None, None]
for role, sl in (("compute waves", slice(0, 8)), ("loader waves", slice(8, 12))):
    s = allw[:, sl]
    nst = int((s[0, 0] > 0).sum())
    rel = (s - t0).astype(np.float64)      # per workgroup: relative to its first wave's start
    if role[0] == "c":
        names = ["start", "issued"] + sum([[f"s{i}.arrive", f"s{i}.issued", f"s{i}.computed", f"s{i}.barrier"] for i in range((nst - 3) // 4)], []) + ["end"]
    else:
        names = ["start", "prologue"] + sum([[f"s{i}.a", f"s{i}.b", f"s{i}.c", f"s{i}.barrier"] for i in range((nst - 3) // 4 + 1)], [])   # v1: a landed, b stored, c issued; v2: a issued(s+2), b = c parked(s+1)
    span = rel[:, :, nst - 1].max(axis=1)
    print(role, "workgroups", s.shape[0], f"; s_memtime ticks (10 ns); per-workgroup span mean {span.mean():.0f} min {span.min():.0f} max {span.max():.0f}")
    prev = None
    for i in range(nst):
        col = rel[:, :, i]
        d = "" if prev is None else f"  (+{col.mean() - prev:7.0f})"
        nm = names[i] if i < len(names) else f"#{i}"
        print(f"{nm:12s} mean {col.mean():8.0f}  min {col.min():6.0f}  max {col.max():6.0f}{d}")
        prev = col.mean()
