"""Per-tensor gradient errors of the full-resolution model on 2 images against the oracle (not a pytest file)."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import vision_amd
from vision_amd.presets import get_preset
from oracle import restatement as R

DEV = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = dict(get_preset("butterfly")["model_config"], dropout=0.0)
sd = R.init_state_like_torch(cfg, cfg["manual_seed"])
model = vision_amd.get_model("ddpm", cfg, compute_dtype=mode).to(DEV)
model.load_state_dict(sd)
model.train()
P_ = cfg["patch_size"]
g = torch.Generator().manual_seed(3)
x = torch.rand((B, 3, P_, P_), generator=g)
eps = torch.randn((B, 3, P_, P_), generator=g)
t = torch.rand((B, 1, 1, 1), generator=g)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
ref_loss64, ref64 = R.loss_and_grads(sd64, cfg, x.double(), eps.double(), t.double())
ref_loss, ref32 = R.loss_and_grads({k: v.clone() for k, v in sd.items()}, cfg, x, eps, t)
loss = model.p_loss(x.to(DEV), noises=eps.to(DEV), diffusion_times=t.to(DEV))
loss.backward()
print("loss hip %.7f oracle32 %.7f oracle64 %.7f" % (float(loss.detach()), float(ref_loss), float(ref_loss64)))
rows = []
num = den = num32 = 0.0
for n, p in model.named_parameters():
    gt = p.grad.detach().double().cpu()
    w64 = ref64[n].double()
    w32 = ref32[n].double()
    e = (gt - w64).norm().item()
    e32 = (w32 - w64).norm().item()
    d = w64.norm().item()
    num += e * e; den += d * d; num32 += e32 * e32
    rows.append((e / (d + 1e-30), e32 / (d + 1e-30), d, n))
print("global: hip vs fp64 %.3e   oracle fp32 vs fp64 %.3e" % ((num / den) ** 0.5, (num32 / den) ** 0.5))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.3e  (oracle32 %.3e)  |g|=%.3e  %s" % r)
