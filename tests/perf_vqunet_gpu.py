"""Throughput of the VQ-VAE-UNet first stage at the sizes of reference configs/butterfly_vqvaeunet.yaml
(224 x 224, batch 16 by default, 2048 codes of 256 features): training step (forward + backward + Adam) and
eval-mode reconstruction; and of latent diffusion over it (configs/butterfly_ldm.yaml: a 256-wide UNet on 7 x 7 maps,
batch 64): training step on latents, on images (first-stage encoder included) and the sampler.  Images/s from HIP events
around whole steps.  Run on the GPU box:

    python tests/perf_vqunet_gpu.py [batch] [steps] [ldm batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vision_amd                         # noqa: E402

CFG = dict(name="vqvaeunet", learning_rate=1e-3, manual_seed=42, h_dims=[3, 64, 64, 128, 128, 256, 256],
           down_index=[0, 1, 0, 1, 1, 1, 1], patch_size=224, sampling_period=1, embed_dim=256, num_embed=2048,
           attn_res=[7], weight_decay=1e-3, scheduler_gamma=0.99)


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    for name, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        torch.manual_seed(42)
        m = vision_amd.get_model("vqvaeunet", CFG, compute_dtype=dtype).to(dev).train()
        x = torch.rand((B, 3, 224, 224), generator=torch.Generator().manual_seed(0)).to(dev)
        opt = m.configure_optimizers()[0][0]

        def step():
            loss = m.training_step((x, None), 0)
            loss.backward()
            opt.step()

        ms = timed(step, steps)
        print(f"{name} train  B={B}: {ms:8.3f} ms/step  {B / ms * 1e3:9.1f} img/s", flush=True)
        m.eval()
        with torch.no_grad():
            ms = timed(lambda: m(x), steps)
        print(f"{name} recon  B={B}: {ms:8.3f} ms/step  {B / ms * 1e3:9.1f} img/s", flush=True)


    LB = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    ldm_cfg = dict(name="ldm", learning_rate=1e-4, manual_seed=42, h_dims=[256, 256, 256], down_index=[0, 0, 0],
                   attn_res=[7], in_res=7, with_conv=True, dropout=0.1, diffusion_steps=50, sampling_period=20, grid=2,
                   weight_decay=1e-5, scheduler_gamma=0.99)
    torch.manual_seed(42)
    fs = vision_amd.get_model("vqvaeunet", CFG, compute_dtype=torch.bfloat16).to(dev).eval()
    m = vision_amd.get_model("ldm", ldm_cfg, compute_dtype=torch.bfloat16).to(dev).train()
    opt = m.configure_optimizers()[0][0]
    z = torch.randn((LB, 256, 7, 7), generator=torch.Generator().manual_seed(0)).to(dev)
    x = torch.rand((LB, 3, 224, 224), generator=torch.Generator().manual_seed(0)).to(dev)

    def lstep(batch):
        loss = m.training_step((batch, None), 0)
        loss.backward()
        opt.step()

    ms = timed(lambda: lstep(z), steps)
    print(f"ldm bf16 train on latents B={LB}: {ms:8.3f} ms/step  {LB / ms * 1e3:9.1f} img/s", flush=True)
    m.set_first_stage(fs)
    ms = timed(lambda: lstep(x), steps)
    print(f"ldm bf16 train on images  B={LB}: {ms:8.3f} ms/step  {LB / ms * 1e3:9.1f} img/s", flush=True)
    m.eval()
    noise = torch.randn((LB, 256, 7, 7), generator=torch.Generator().manual_seed(1)).to(dev)
    ms = timed(lambda: m.generate(LB, 10, initial_noise=noise), 3, warmup=1)
    print(f"ldm bf16 sampler 10 steps B={LB}: {ms:8.3f} ms  {LB * 10 / ms * 1e3:9.1f} latent-steps/s", flush=True)
    ms = timed(lambda: m.decode_latents(noise), 5, warmup=1)
    print(f"ldm bf16 decode B={LB}: {ms:8.3f} ms  {LB / ms * 1e3:9.1f} img/s", flush=True)


if __name__ == "__main__":
    main()
