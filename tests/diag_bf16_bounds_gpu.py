"""Diagnostics (not collected): measured bf16 gradient errors of the engine beside the reference's own bf16 autocast
errors, per golden configuration -- the numbers the bf16 bounds of tests/test_model_gpu.py are set from."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, sub
import test_model_gpu as T
DEV = torch.device("cuda:0")
for name in ("t0", "t1", "t0_drop", "t0_pool"):
    g, cfg, B = load_golden(name)
    pred64, g64, e32, ebf, ebf_all, e32_all = T.exact_gradients(name, g, cfg)
    for mode in ("bf16-direct", "bf16-auto"):
        m = T.make_model(cfg, sub(g, "state0"), mode).train()
        m.set_dropout_masks(T.masks_for(g, cfg, torch.bfloat16))
        out = m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV))
        loss = torch.nn.functional.l1_loss(out, g["eps"].to(DEV)); loss.backward()
        grads = {k: p.grad for k, p in m.named_parameters()}
        errs = {k: T.rel(grads[k], g64[k]) for k in e32}
        keys = sorted(e32)
        e_all = T.rel(T._cat(grads, keys), T._cat(g64, keys))
        worst = max(errs, key=errs.get)
        ratio = max(errs[k] / max(ebf[k], 1e-9) for k in errs)
        print(f"{name:8s} {mode:12s} fwd {T.rel(out, pred64):.2e}  worst tensor {errs[worst]:.3f} ({worst}; reference bf16 {ebf[worst]:.3f})  "
              f"max mine/reference {ratio:.2f}  all {e_all:.3f} (reference {ebf_all:.3f})", flush=True)
