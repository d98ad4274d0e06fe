"""TEST INFRASTRUCTURE ONLY -- loads the *unmodified* reference `models/ddpm.py`.

Used by `oracle/make_golden.py` in the build container, where `/root/reference` is
mounted, to generate the committed fixtures under `tests/golden/`.  It never runs
on the GPU box (the reference does not travel) and is never imported by the
product package `vision_amd`.

The reference file imports three packages that are absent from this image
(`pytorch_lightning`, `torchmetrics`, `torchvision`, reference models/ddpm.py:9-11).
They are replaced by in-memory stand-ins registered in `sys.modules` before the
file is executed; none of them takes part in the arithmetic of the hot path:

* `pytorch_lightning.LightningModule` -> an `nn.Module` with a no-op `log`, a
  `current_epoch` of 0 and a writable `device` attribute (the reference assigns
  `self.device = x.device` in `forward`, models/ddpm.py:209, and reads it in
  `diffusion_schedule`, models/ddpm.py:245).
* `torchmetrics.Accuracy` -> unused placeholder.
* `torchvision.utils.save_image` -> unused placeholder.
"""
import importlib.util
import os
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("VISION_REFERENCE_ROOT", "/root/reference")


def _install_stubs():
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(nn.Module):
            def __init__(self, *a, **k):
                super().__init__()
                object.__setattr__(self, "_dev", torch.device("cpu"))
                self.current_epoch = 0

            @property
            def device(self):
                return self._dev

            @device.setter
            def device(self, value):
                object.__setattr__(self, "_dev", value)

            def log(self, *a, **k):
                return None

        pl.LightningModule = LightningModule
        sys.modules["pytorch_lightning"] = pl
    if "torchmetrics" not in sys.modules:
        tm = types.ModuleType("torchmetrics")
        tm.Accuracy = object
        sys.modules["torchmetrics"] = tm
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.save_image = lambda *a, **k: None
        tv.utils = tvu
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tvu


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "ddpm.py"))


def load_reference_ddpm():
    """Return the reference module object (its `DDPMModel`, `timeembed`, ...)."""
    if not reference_available():
        raise FileNotFoundError(
            f"reference not mounted at {REFERENCE_ROOT}; golden vectors can only be "
            "regenerated in the build container")
    _install_stubs()
    path = os.path.join(REFERENCE_ROOT, "models", "ddpm.py")
    spec = importlib.util.spec_from_file_location("ref_ddpm", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _load_models_file(stem):
    """Execute reference models/<stem>.py as a submodule of an empty stand-in package whose search path is the
    reference's models/ directory (the files do `from . import Custumnn`); models/__init__.py itself, which imports
    every model family, is not run."""
    if not reference_available():
        raise FileNotFoundError(f"reference not mounted at {REFERENCE_ROOT}")
    _install_stubs()
    pkg_dir = os.path.join(REFERENCE_ROOT, "models")
    if "ref_models" not in sys.modules:
        pkg = types.ModuleType("ref_models")
        pkg.__path__ = [pkg_dir]
        sys.modules["ref_models"] = pkg
    spec = importlib.util.spec_from_file_location(f"ref_models.{stem}", os.path.join(pkg_dir, f"{stem}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"ref_models.{stem}"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_vqblock():
    """The reference module models/vqvaeunet.py (its `VQBlock`, models/vqvaeunet.py:104-146, and `VQVAEUNETModel`)."""
    return _load_models_file("vqvaeunet")


def load_reference_ldm():
    """The reference module models/ldm.py (its blocks and `LatentDiffusion`)."""
    return _load_models_file("ldm")
