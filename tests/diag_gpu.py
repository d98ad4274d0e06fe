"""Diagnostics (not collected by pytest): per-layer forward error of the HIP engine against the
reference taps of tests/golden/t0.safetensors, and gradient errors against fp64 truth."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, sub, noise_keys
import vision_amd
from vision_amd import _lib
from oracle import restatement as R

DEV = torch.device("cuda:0")
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()

for name in sys.argv[1:] or ["t0", "t1"]:
    g, cfg, B = load_golden(name)
    sd = sub(g, "state0")
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    taps64 = {}
    n2 = (g["sched_n"] ** 2).double()
    taps64["pred_noises"] = R.unet_forward({k: v.clone() for k, v in sd64.items()}, cfg, g["x_t"].double(), n2, True, None, taps64)
    leaves0 = {k: v.detach().clone().requires_grad_(True) for k, v in sd64.items() if R.is_param(k)}
    work0 = {k: v.clone() for k, v in sd64.items()}; work0.update(leaves0)
    l0 = torch.nn.functional.l1_loss(R.unet_forward(work0, cfg, g["x_t"].double(), n2, True), g["eps"].double())
    g64 = dict(zip(leaves0, torch.autograd.grad(l0, list(leaves0.values()))))
    ref = sub(g, "grad")
    for mode, (dtype, algo) in {"f32-direct": (torch.float32, 1), "f32-auto": (torch.float32, 0), "bf16-auto": (torch.bfloat16, 0)}.items():
        m = vision_amd.get_model("ddpm", cfg, compute_dtype=dtype, kernel_algo=algo)
        m.load_state_dict(sd); m = m.to(DEV).train()
        eng = m._ensure_engine(); eng.debug_taps = {}
        out = m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV))      # identical x_t / n^2 bits as the reference
        loss = torch.nn.functional.l1_loss(out, g["eps"].to(DEV))
        loss.backward()
        m._last_pred_nhwc = out.permute(0, 2, 3, 1).detach()
        print(f"== {name} {mode}: loss {loss.item():.8f} ref {g['loss'].item():.8f}")
        for k, (h, res2) in eng.debug_taps.items():
            hh = h.float()
            if res2 is not None:
                hh = hh - res2.float()
            t64 = taps64[k]
            tref = g.get("tap/" + k)
            print(f"   fwd {k:18s} err vs fp64 {rel(hh.permute(0,3,1,2), t64):.2e}" + (f"   (reference fp32: {rel(tref, t64):.2e})" if tref is not None else ""))
        print(f"   pred err {rel(m._last_pred_nhwc.float().permute(0,3,1,2), taps64['pred_noises']):.2e}")
        # exact gradient of the same function with MY sign pattern (isolates L1 sign flips)
        pred_mine = m._last_pred_nhwc.float().permute(0, 3, 1, 2).cpu().double()
        sgn = torch.sign(pred_mine - g["eps"].double())
        sgn64 = torch.sign(taps64["pred_noises"] - g["eps"].double())
        print("   sign flips vs fp64:", int((sgn != sgn64).sum()), "of", sgn.numel(),
              "| reference fp32 flips:", int((torch.sign(g["fwd_train"].double() - g["eps"].double()) != sgn64).sum()) if "fwd_train" in g else "n/a")
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd64.items() if R.is_param(k)}
        work = dict(sd64); work.update(leaves)
        out = R.unet_forward({k: (v.clone() if not R.is_param(k) else v) for k, v in work.items()}, cfg, g["x_t"].double(), n2, True)
        gs = torch.autograd.grad((out * sgn).sum() / sgn.numel(), [leaves[k] for k in leaves])
        gflip = dict(zip(leaves, gs))
        rows2 = sorted(((rel(p.grad, gflip[k]), k) for k, p in m.named_parameters() if k not in noise_keys(g)), reverse=True)
        print("   worst grad errors vs fp64 gradient with my sign pattern:", [(k, "%.1e" % e) for e, k in rows2[:6]])
        rows = []
        for k, p in m.named_parameters():
            if k in noise_keys(g): continue
            rows.append((rel(p.grad, g64[k]) / max(rel(ref[k], g64[k]), 1e-7), rel(p.grad, g64[k]), rel(ref[k], g64[k]), k))
        rows.sort(reverse=True)
        for r in rows[:12]:
            print("   grad %-42s mine %.2e  ref32 %.2e  ratio %.1f" % (r[3], r[1], r[2], r[0]))
