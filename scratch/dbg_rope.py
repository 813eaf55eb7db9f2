import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
DEV="cuda:0"
dt = torch.float16
g = torch.Generator().manual_seed(3)
for (B, H, L, D, dims, off, scale) in ((2, 8, 17, 128, 128, 0, 1.0), (2, 6, 300, 64, 64, 2048, 1.0)):
    x = torch.randn(B, L, H, D, generator=g).to(dt)
    inv = (1.0 / (10000.0 ** (torch.arange(0, dims, 2, dtype=torch.float32) / dims))).to(DEV)
    xd = x.to(DEV).transpose(1, 2)
    got = ops.rope(xd, inv, off, scale, False)
    for t in (0, 7, L - 1):
        ref = ops.rope(xd[:, :, t:t + 1], inv, off + t, scale, False)
        a = got[:, :, t:t+1]
        d = (a != ref)
        idx = d.nonzero()
        print((B,H,L,D), t, int(d.sum()), idx[:6].tolist(), [(a[tuple(i)].item(), ref[tuple(i)].item(), xd[:, :, t:t+1][tuple(i)].item()) for i in idx[:3]])
        print("  nan", torch.isnan(a).sum().item(), torch.isnan(ref).sum().item())
