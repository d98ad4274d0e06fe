"""Host-side helpers around the hot path: PNG grid writer (torchvision is not required) and the
safetensors checkpoint callback of the reference's utils.py."""
import os
import shutil

import numpy as np
import torch


def make_grid(images: torch.Tensor, nrow: int, padding: int = 2, normalize: bool = False) -> torch.Tensor:
    """[N,C,H,W] -> [C, rows*(H+pad)+pad, cols*(W+pad)+pad]; min-max normalisation over the whole
    batch when `normalize` (what torchvision.utils.save_image(normalize=True) does, reference
    models/ddpm.py:362-366)."""
    x = images.detach().float().cpu()
    if normalize:
        lo, hi = float(x.min()), float(x.max())
        x = (x.clamp(lo, hi) - lo) / max(hi - lo, 1e-5)
    n, c, h, w = x.shape
    cols = min(nrow, n)
    rows = int(np.ceil(n / cols))
    grid = torch.zeros((c, rows * (h + padding) + padding, cols * (w + padding) + padding))
    for i in range(n):
        r, q = divmod(i, cols)
        top, left = r * (h + padding) + padding, q * (w + padding) + padding
        grid[:, top:top + h, left:left + w] = x[i]
    return grid


def save_image_grid(images, path, normalize=True, nrow=8):
    from PIL import Image
    grid = make_grid(images, nrow=nrow, normalize=normalize)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    if arr.shape[2] == 1:
        arr = arr[:, :, 0]
    Image.fromarray(arr).save(path)


class SafetensorsCheckpoint:
    """Top-k + last `.safetensors` weight checkpoints keyed on a monitored metric; behaviour of the
    reference's utils.py:7-95 (weights only: no optimiser / scheduler / epoch state)."""

    def __init__(self, config_path, dirpath, filename="{epoch}-{step}-{val_loss:.2f}", monitor="val_loss",
                 mode="min", save_top_k=1, save_last=False, every_n_epochs=1):
        self.config_path, self.dirpath, self.filename = config_path, dirpath, filename
        self.monitor, self.mode, self.save_top_k = monitor, mode, save_top_k
        self.save_last, self.every_n_epochs = save_last, every_n_epochs
        self.best_k_models = {}
        self.kth_best_model_path = ""
        self.best_model_score = None
        self.best_model_path = ""
        self.last_model_path = ""
        self.kth_value = float("inf") if mode == "min" else -float("inf")

    def _path_for(self, epoch, step, metrics):
        name = self.filename.format(epoch=epoch, step=step, **metrics)
        return os.path.join(self.dirpath, f"{name}.safetensors")

    def _save(self, trainer, filepath):
        from safetensors.torch import save_file
        os.makedirs(os.path.dirname(filepath), exist_ok=True)
        state = {k: v.detach().contiguous().cpu() for k, v in trainer.lightning_module.state_dict().items()}
        save_file(state, filepath)

    def _better(self, a, b):
        return a < b if self.mode == "min" else a > b

    def _update_best_and_save(self, current, filepath, trainer):
        if self.save_top_k == 0:
            return
        if len(self.best_k_models) < self.save_top_k:
            self.best_k_models[filepath] = current
            self._save(trainer, filepath)
        elif self._better(current, self.kth_value):
            old = self.kth_best_model_path
            self.best_k_models.pop(old, None)
            if old and os.path.exists(old):
                os.remove(old)
            self.best_k_models[filepath] = current
            self._save(trainer, filepath)
        pick_worst, pick_best = (max, min) if self.mode == "min" else (min, max)
        self.kth_best_model_path = pick_worst(self.best_k_models, key=self.best_k_models.get)
        self.kth_value = self.best_k_models[self.kth_best_model_path]
        self.best_model_path = pick_best(self.best_k_models, key=self.best_k_models.get)
        self.best_model_score = self.best_k_models[self.best_model_path]

    def on_validation_end(self, trainer, pl_module):
        if trainer.current_epoch % self.every_n_epochs != 0:
            return
        metrics = trainer.callback_metrics
        value = metrics.get(self.monitor)
        if value is None:
            return
        value = float(value)
        self._update_best_and_save(value, self._path_for(trainer.current_epoch, trainer.global_step,
                                                         {k: float(v) for k, v in metrics.items()}), trainer)
        if self.save_last:
            last = os.path.join(self.dirpath, "last.safetensors")
            self._save(trainer, last)
            self.last_model_path = last

    def on_train_start(self, trainer, pl_module):
        os.makedirs(self.dirpath, exist_ok=True)
        shutil.copy(self.config_path, os.path.join(self.dirpath, "../config.yaml"))
