"""diagnostic: the 3-bit matrix-core fused MLP launch forced on chosen layers of the S22 mix (MX3_LAYERS=1,2 / 30 / 1,2,30 / none)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gbx_lm_amd import synthetic as S, decode
chosen = [int(v) for v in os.environ.get("MX3_LAYERS", "").split(",") if v]
orig = decode.FusedDecodeRunner._fused_descs
def patched(self, li, *a, **k):
    self._mostly_3bit = li in chosen
    return orig(self, li, *a, **k)
decode.FusedDecodeRunner._fused_descs = patched
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
r = [bench.decode_leg(model, cfg, dev, 1, 128, 20, 5)["tok_s"] for _ in range(3)]
print(f"MX3 on layers {chosen}: " + " ".join(f"{x:.1f}" for x in r))
