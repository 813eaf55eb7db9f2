"""Host-side cost per call of the Python wrappers (no device sync inside the timed loops): what bounds eager prefill of
short prompts."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, synthetic as S
from gbx_lm_amd.models.cache import make_prompt_cache
dev = 'cuda:0'
n, k, b, g, m = 4096, 4096, 2, 128, 128
qw = torch.randint(0, 2**31 - 1, (n, k // 32 * b), dtype=torch.int32, device=dev).view(torch.uint32)
sc = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16); ze = (-sc.float() * 1.5).to(torch.bfloat16)
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = torch.ones(k, device=dev, dtype=torch.bfloat16)
def t(f, reps=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / reps * 1e6
print(f"ops.quantized_matmul host us/call: {t(lambda: ops.quantized_matmul(x, qw, sc, ze, group_size=g, bits=b)):.1f}")
print(f"ops.rmsnorm host us/call:          {t(lambda: ops.rmsnorm(x, w, 1e-5)):.1f}")
print(f"torch add host us/call:            {t(lambda: x + x):.1f}")
print(f"ops.swiglu host us/call:           {t(lambda: ops.swiglu(x, x)):.1f}")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy('S22', cfg['num_hidden_layers'], 128), device=dev, seed=0)
toks = (torch.arange(128, device=dev) % 1000)[None]
def fwd():
    with torch.no_grad():
        model.model(toks, cache=make_prompt_cache(model))
print(f"128-token prefill forward, host us/call: {t(fwd, 10):.0f}")
torch.cuda.synchronize(); t0 = time.perf_counter(); fwd(); torch.cuda.synchronize(); print(f"  wall incl. device: {(time.perf_counter()-t0)*1e3:.2f} ms")
