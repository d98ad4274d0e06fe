"""`pytorch_lightning` is optional: when it is importable DDPMModel subclasses the real
LightningModule; otherwise this minimal stand-in supplies the attributes the reference's
`models/ddpm.py` touches (`log`, `device`, `trainer`, `logger`, `current_epoch`)."""
import torch
from torch import nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # ImportError, or a broken install
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def __init__(self, *args, **kwargs):
            super().__init__()
            object.__setattr__(self, "_device_hint", None)
            self.trainer = None
            self.logger = None
            self.logged = {}

        # The reference assigns `self.device = x.device` in forward (models/ddpm.py:209); under
        # Lightning `device` is a property.  Accept the assignment and otherwise report where the
        # parameters live.
        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return self._device_hint or torch.device("cpu")

        @device.setter
        def device(self, value):
            object.__setattr__(self, "_device_hint", value)

        @property
        def current_epoch(self):
            return self.trainer.current_epoch if self.trainer is not None else 0

        @property
        def global_rank(self):
            return self.trainer.global_rank if self.trainer is not None else 0

        def log(self, name, value, *args, **kwargs):
            if self.trainer is not None and hasattr(self.trainer, "record"):
                self.trainer.record(name, value, **kwargs)
            else:
                self.logged[name] = value
