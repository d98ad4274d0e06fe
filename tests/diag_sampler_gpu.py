"""Kernel-level view of the sampler (not a pytest file): run under
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/samp -- python tests/diag_sampler_gpu.py
to see where a 16-image denoising step spends its time."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import vision_amd                                        # noqa: E402
from vision_amd.presets import get_preset                 # noqa: E402

cfg = get_preset("butterfly")["model_config"]
model = vision_amd.get_model("ddpm", cfg, compute_dtype="bf16").to("cuda:0").eval()
n, steps = 16, 10
with torch.no_grad():
    model.generate(n, steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        model.generate(n, steps)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 3
print("%.3f ms per denoising step of %d images = %.0f image-steps/s" % (el / steps * 1e3, n, n * steps / el))
