"""Forward error of the full-resolution model (train-mode BN) against the fp64 oracle (not a pytest file)."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import vision_amd
from vision_amd.presets import get_preset
from oracle import restatement as R

DEV = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = 2
cfg = dict(get_preset("butterfly")["model_config"], dropout=0.0)
sd = R.init_state_like_torch(cfg, cfg["manual_seed"])
model = vision_amd.get_model("ddpm", cfg, compute_dtype=mode).to(DEV)
model.load_state_dict(sd)
model.train()
P_ = cfg["patch_size"]
g = torch.Generator().manual_seed(3)
x = torch.randn((B, 3, P_, P_), generator=g)
t = torch.rand((B, 1, 1, 1), generator=g)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
taps64, taps32 = {}, {}
with torch.no_grad():
    p64 = R.unet_forward(sd64, cfg, x.double(), t.double(), training=True, taps=taps64)
    p32 = R.unet_forward({k: v.clone() for k, v in sd.items()}, cfg, x, t, training=True, taps=taps32)
    eng = model._ensure_engine()
    eng.debug_taps = {}
    ph = model(x.to(DEV), t.to(DEV)).double().cpu()
rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
print("pred: hip %.3e   oracle32 %.3e" % (rel(ph, p64), rel(p32, p64)))
for k in taps64:
    if k in eng.debug_taps and k in taps32:
        out, skip = eng.debug_taps[k]
        h = out.double().cpu()
        if h.dim() == 4 and h.shape != taps64[k].shape:
            h = h.permute(0, 3, 1, 2)
        if h.shape == taps64[k].shape:
            print("%-22s hip %.3e  oracle32 %.3e" % (k, rel(h, taps64[k]), rel(taps32[k], taps64[k])))
