"""Does the RELATIVE alignment of a block's weight tensors (gate vs up vs down rows streamed together) change the token rate?
build_model allocates them back to back: gate, up (exactly 14 x 2 MiB later), down ...  Here up / down are re-allocated behind small pads."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gbx_lm_amd import synthetic as S
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
def rate():
    model._decode_runner = None
    return [round(bench.decode_leg(model, cfg, dev, 1, 128, 20, 5)["tok_s"], 1) for _ in range(2)]
L0 = model.model.layers[5].mlp
print("gate/up/down data_ptr deltas (layer 5):", (L0.up_proj.qweight.data_ptr() - L0.gate_proj.qweight.data_ptr()) / 2**20, (L0.down_proj.qweight.data_ptr() - L0.up_proj.qweight.data_ptr()) / 2**20, "MiB", flush=True)
print("as built:", rate(), flush=True)
keep = []
for off_kb in (0, 1, 4, 32, 68, 516, 1100):
    for blk in model.model.layers:
        ups, dns = blk.mlp.up_proj, blk.mlp.down_proj
        nb = ups.qweight.numel() * 4
        arena = torch.empty((2 * nb + (4 << 20),), dtype=torch.uint8, device=dev)      # its own 2 MiB-aligned segment
        keep.append(arena)
        o1 = off_kb * 1024
        o2 = ((nb + 2 * off_kb * 1024 + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + 2 * off_kb * 1024
        for m, o in ((ups, o1), (dns, o2)):
            dst = arena[o:o + nb].view(torch.uint32).view(m.qweight.shape)
            dst.copy_(m.qweight)
            m._buffers["qweight"] = dst
            m.__dict__.pop("_gbx_companions", None)
    torch.cuda.synchronize()
    d = (L0.up_proj.qweight.data_ptr() - L0.gate_proj.qweight.data_ptr())
    d2 = (L0.down_proj.qweight.data_ptr() - L0.gate_proj.qweight.data_ptr())
    print(f"up at +{off_kb} KB, down at +{2 * off_kb} KB (mod 2 MiB vs gate: {(d % (2 << 20)) / 1024:.0f} / {(d2 % (2 << 20)) / 1024:.0f} KB): ", rate(), flush=True)
