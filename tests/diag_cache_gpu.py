"""Diagnostics (not collected): how often the eval-mode BatchNorm coefficients are recomputed by a sampler call."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import vision_amd
from vision_amd import engine as E
from vision_amd.presets import get_preset
cfg = get_preset("butterfly")["model_config"]
model = vision_amd.get_model("ddpm", cfg, compute_dtype="bf16").to("cuda:0").eval()
count = {}
orig = E.call
def counting(name, *a):
    count[name] = count.get(name, 0) + 1
    return orig(name, *a)
E.call = counting
eng = model._ensure_engine()
with torch.no_grad():
    for i in range(2):
        count.clear()
        v0 = (eng.coef_epoch, eng.flat_b._version)
        model.generate(16, 10)
        print("call", i, "key before/after", v0, (eng.coef_epoch, eng.flat_b._version), {k: v for k, v in count.items() if "finalize" in k or "linear" in k or "repack" in k})
