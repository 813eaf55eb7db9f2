"""cfg4: prefill of Llama-3-8B shapes (all 4-bit g=128 'S40', or S22) through the model's own forward: batch x seq tokens,
chunked at 2048 like generate_step.  Reports tokens/s, the time inside the QuantizedLinear GEMMs and their MFMA rate."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import synthetic
from gbx_lm_amd.models.cache import make_prompt_cache
strategy = sys.argv[1] if len(sys.argv) > 1 else "S40"
B, T = int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev = torch.device("cuda:0")
cfg = synthetic.LLAMA3_8B
model = synthetic.build_model(cfg, synthetic.make_strategy(strategy, cfg["num_hidden_layers"], 128), device=dev, seed=0)
toks = (torch.arange(B * T, device=dev) % cfg["vocab_size"]).view(B, T)
def run():
    cache = make_prompt_cache(model)
    with torch.no_grad():
        for s in range(0, T, 2048):
            h = model.model(toks[:, s:s + 2048], cache=cache)
    return h
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
lin_flops = 2.0 * B * T * sum(m.input_dims * m.output_dims for m in model.modules() if m.__class__.__name__ == "QuantizedLinear")
# GEMM-only time: the seven projections of every layer at M = B*2048 rows, twice (two chunks of 2048)
from gbx_lm_amd import ops
x = {}
tg = 0.0
for name in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"):
    for blk in model.model.layers:
        m = getattr(blk.self_attn, name, None) or getattr(blk.mlp, name)
        k = m.input_dims
        if k not in x: x[k] = torch.randn(B * 2048, k, device=dev).to(torch.bfloat16)
        m(x[k])
torch.cuda.synchronize()
e0.record()
for rep in range(T // 2048):
    for blk in model.model.layers:
        for m in (blk.self_attn.q_proj, blk.self_attn.k_proj, blk.self_attn.v_proj, blk.self_attn.o_proj, blk.mlp.gate_proj, blk.mlp.up_proj, blk.mlp.down_proj):
            m(x[m.input_dims])
e1.record(); torch.cuda.synchronize()
tg = e0.elapsed_time(e1)
print(f"prefill {strategy} B={B} T={T}: {ms:.1f} ms = {B*T/ms*1e3:,.0f} tok/s; QuantizedLinear GEMMs alone {tg:.1f} ms = {lin_flops/tg/1e9:.0f} TFLOP/s "
      f"({lin_flops/tg/1e9/2500*100:.1f} % of the 2.5 PFLOP/s dense bf16 peak); linear flops {lin_flops:.3e}")
