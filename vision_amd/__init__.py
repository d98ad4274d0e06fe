"""MI355X-native DDPM UNet training step and sampler, the VQ-VAE-UNet first stage and latent diffusion over it, behind
the reference's model interface.

`get_model('ddpm' | 'vqvaeunet', model_config)` mirrors the reference registry (models/__init__.py:13-26); 'ldm' is
this package's name for the reference's unregistered models/ldm.py.
Importing this package loads libvision_amd.so; there is no CPU or eager fallback."""
from . import _lib  # noqa: F401  (fails loudly when the HIP extension has not been built)
from .ddpm import DDPMModel
from .ldm import LatentDiffusion
from .vqvaeunet import VQVAEUNETModel

chose_model = {"ddpm": DDPMModel, "vqvaeunet": VQVAEUNETModel, "ldm": LatentDiffusion}


def get_model(name, config, **kwargs):
    return chose_model[name](config, **kwargs)


__all__ = ["DDPMModel", "VQVAEUNETModel", "LatentDiffusion", "get_model", "chose_model"]
