"""Throughput of the data module's two transform paths on synthetic JPEG files (not a pytest file).

    python tests/perf_input_gpu.py [n_images] [height] [width] [workers]
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from vision_amd.trainer import ImageFolderDataModule   # noqa: E402

n, h, w, workers = (int(a) for a in (sys.argv[1:] + ["512", "500", "667", "16"])[:4])
DEV = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as root:
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(root, "c"))
    base = rng.integers(0, 256, (h // 8, w // 8, 3), dtype=np.uint8)
    for i in range(n):
        img = np.roll(base, i, axis=1).repeat(8, 0).repeat(8, 1)
        Image.fromarray(img, "RGB").save(os.path.join(root, "c", f"{i:05d}.jpg"), quality=90)
    for mode, nw in ((True, workers), (True, 4), (False, 1)):
        dm = ImageFolderDataModule(root, patch_size=224, batch_size=64, use_manual_split=True, val_split=0.0, test_split=0.0,
                                   device=DEV, gpu_transform=mode, num_workers=nw)
        dm.setup()
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            cnt = 0
            for x, y in dm.train_dataloader():
                cnt += x.shape[0]
            torch.cuda.synchronize()
            dt = time.time() - t0
        print(f"{'GPU transform, %d decode threads' % nw if mode else 'host transform (PIL), 1 thread'}: "
              f"{cnt / dt:8.1f} images/s ({h}x{w} JPEG -> 224x224 fp32)", flush=True)
