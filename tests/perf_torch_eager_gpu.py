"""What the reference's own formulation reaches on the same GPU (not a pytest file, not a product path).

    python tests/perf_torch_eager_gpu.py [batch] [steps]

Runs the oracle's plain-PyTorch restatement of the training step (reference models/ddpm.py:208-269,
:311-352: forward, L1 loss, autograd backward, torch.optim.AdamW) on cuda:0 with the butterfly
model at 224 x 224, the way the reference's `precision: 16-mixed` trainer would: torch.autocast
to bf16 around the forward (fp16 + GradScaler on the reference; bf16 needs no scaler), fp32
master weights, channels-last off (the reference does not set it).  Convolutions go to MIOpen,
everything else to torch's eager kernels.  Prints one JSON line; the number is context for the
bench line of the HIP path, measured on the same box in the same call.
"""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from oracle import restatement as R                   # noqa: E402
from vision_amd.presets import get_preset              # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    cfg = get_preset("butterfly")["model_config"]
    sd = {k: v.to(dev) for k, v in R.init_state_like_torch(cfg, cfg["manual_seed"]).items()}
    names = [k for k in sd if R.is_param(k)]
    params = [sd[k].requires_grad_(True) for k in names]
    opt = torch.optim.AdamW(params, lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"])
    P = cfg["patch_size"]
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand((B, 3, P, P), generator=g, device=dev)

    def step():
        eps = torch.randn((B, 3, P, P), generator=g, device=dev)
        t = torch.rand((B, 1, 1, 1), generator=g, device=dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = R.p_loss(sd, cfg, x, eps, t, True)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"what": "oracle restatement, torch eager + MIOpen, bf16 autocast, butterfly 224x224",
                      "batch": B, "steps": steps, "ms_per_step": round(el / steps * 1e3, 2),
                      "images_per_s": round(B * steps / el, 1), "loss": round(float(loss.detach()), 5),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))


if __name__ == "__main__":
    main()
