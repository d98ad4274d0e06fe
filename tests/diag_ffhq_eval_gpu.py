"""Per-module error of the small ffhq wiring against the fp64 oracle, train or eval mode (not a pytest file).
python tests/diag_ffhq_eval_gpu.py [fp32|bf16] [eval|train]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import vision_amd
from vision_amd.presets import get_preset
from oracle import restatement as R

DEV = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
training = (sys.argv[2] if len(sys.argv) > 2 else "eval") == "train"
cfg = dict(get_preset("ffhq")["model_config"], patch_size=128, attn_res=[2, 4], latent_dim=[3, 128, 128], dropout=0.0)
sd = R.init_state_like_torch(cfg, cfg["manual_seed"])
model = vision_amd.get_model("ddpm", cfg, compute_dtype=mode).to(DEV)
model.load_state_dict(sd)
model.train(training)
g = torch.Generator().manual_seed(3)
x = torch.randn((2, 3, 128, 128), generator=g)
t = torch.rand((2, 1, 1, 1), generator=g)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
taps64 = {}
with torch.no_grad():
    p64 = R.unet_forward(sd64, cfg, x.double(), t.double(), training=training, taps=taps64)
    eng = model._ensure_engine()
    eng.debug_taps = {}
    ph = model(x.to(DEV), t.to(DEV)).double().cpu()
rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
print("pred: hip %.3e" % rel(ph, p64))
for k in taps64:
    if k in eng.debug_taps:
        out, skip = eng.debug_taps[k]
        h = out.double().cpu()
        if skip is not None:
            h = h - skip.double().cpu()
        h = h.permute(0, 3, 1, 2)
        print("%-18s %-5s hip %.3e  max|ref| %.3e  shape %s" % (k, "+skip" if skip is not None else "", rel(h, taps64[k]), float(taps64[k].abs().max()), tuple(h.shape)))
