import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
from gbx_lm_amd import infer_opt
dev='cuda:0'
torch.manual_seed(0)
lg = (torch.randn(128256, device=dev) * 3).to(torch.bfloat16)
for _ in range(3): ops.eminf_optimize(lg)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): b, h = ops.eminf_optimize(lg)
e1.record(); torch.cuda.synchronize()
print(f"device EMINF (V=128256): {e0.elapsed_time(e1)*1e3/20:.1f} us per call, entropies {h.tolist()}")
