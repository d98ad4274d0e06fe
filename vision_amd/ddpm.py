"""MI355X-native `DDPMModel`: the LightningModule surface of the reference's `models/ddpm.py`
(constructor, attributes, method names, state_dict keys, YAML schema) on top of the HIP engine.

Parameters live in ordinary torch containers with the reference's module tree, so
`state_dict()` keys/shapes, default initialisation and its RNG order (`seed_everything` then
construct, reference run.py:44-46) are identical and safetensors checkpoints interchange both
ways.  The containers are never *called*: all arithmetic is done by libvision_amd.so through
`vision_amd.engine.Engine`.
"""
import math
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import call
from .engine import Engine, _ptr, _stream
from .lightning_shim import LightningModule
from .optim import FusedAdamW
from .plan import make_plan
from .utils import save_image_grid

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "bf16-mixed": torch.bfloat16,
           "16-mixed": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32,
           "32": torch.float32, "32-true": torch.float32, 32: torch.float32}


def _conv(cin, cout, k, **kw):
    return nn.Conv2d(cin, cout, kernel_size=(k, k), **kw)


class _ResParams(nn.Module):
    """Parameter holder with the key layout of the reference ResNetBlock (models/ddpm.py:43-68)."""

    def __init__(self, cin, cout, act, time_features, dropout=0.0):
        super().__init__()
        self.linear_temb = _conv(time_features, cout, 1)
        self.block1 = nn.ModuleList([nn.Sequential(nn.BatchNorm2d(cin), act, _conv(cin, cout, 3, padding=1))])
        self.block2 = nn.ModuleList([nn.Sequential(nn.BatchNorm2d(cout), act, nn.Dropout(dropout),
                                                   _conv(cout, cout, 3, padding=1))])
        self.layers = nn.ModuleList([_conv(cin, cout, 1)]) if cin != cout else None


class _AttnParams(nn.Module):
    """reference AttnBlock (models/ddpm.py:89-98)."""

    def __init__(self, c):
        super().__init__()
        self.features = c
        self.norm = nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)
        self.q_proj, self.k_proj, self.v_proj = _conv(c, c, 1), _conv(c, c, 1), _conv(c, c, 1)
        self.linear = _conv(c, c, 1)


class _DownParams(nn.Module):
    def __init__(self, c, with_conv):
        super().__init__()
        self.layers = nn.ModuleList([_conv(c, c, 3, stride=2, padding=1) if with_conv else nn.AvgPool2d((2, 2))])


class _UpParams(nn.Module):
    def __init__(self, c, with_conv):
        super().__init__()
        self.layers = nn.ModuleList([nn.Upsample(scale_factor=2, mode="nearest")])
        if with_conv:
            self.layers.append(_conv(c, c, 3, padding=1))


class _UNetFn(torch.autograd.Function):
    """forward(x, t) as one autograd node: the engine's hand-written backward runs when the
    caller backpropagates through the prediction."""

    @staticmethod
    def forward(ctx, model, x, t, anchor):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("vision_amd computes parameter gradients only: the network input x and the time t "
                               "must not require grad")
        eng = model._ensure_engine()
        B, Cc, H, W = x.shape
        xn = torch.empty((B, H, W, Cc), dtype=eng.dtype, device=x.device)
        call("vamd_nchw_to_nhwc", _ptr(x), _ptr(xn), None, B, Cc, H, W, eng.dt, _stream())
        tape = [] if ctx.needs_input_grad[3] else None     # grad mode is off inside forward()
        pred = eng.forward(xn, t.reshape(B).contiguous().float(), model.training, tape, model._pop_drop_masks())
        out = torch.empty((B, pred.shape[3], H, W), dtype=torch.float32, device=x.device)
        call("vamd_nhwc_to_nchw", _ptr(pred), _ptr(out), B, pred.shape[3], H, W, eng.dt, _stream())
        ctx.eng, ctx.tape = eng, tape
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        B, Cc, H, W = dout.shape
        dn = torch.empty((B, H, W, Cc), dtype=eng.dtype, device=dout.device)
        call("vamd_nchw_to_nhwc", _ptr(dout.contiguous().float()), _ptr(dn), None, B, Cc, H, W, eng.dt, _stream())
        eng.backward(ctx.tape, dn)
        ctx.tape = None
        return None, None, None, None


class _PLossFn(torch.autograd.Function):
    """Fused training objective: schedule + q_sample -> UNet -> L1, one autograd node."""

    @staticmethod
    def forward(ctx, model, images, noises, times, anchor):
        if any(ctx.needs_input_grad[1:4]):
            raise RuntimeError("vision_amd computes parameter gradients only: images, noises and diffusion times "
                               "must not require grad")
        eng = model._ensure_engine()
        B, Cc, H, W = images.shape
        dev = images.device
        xt = torch.empty((B, H, W, Cc), dtype=eng.dtype, device=dev)
        rates = torch.empty((3, B), dtype=torch.float32, device=dev)       # n, s, n^2
        call("vamd_qsample", _ptr(images), _ptr(noises), _ptr(times), _ptr(xt), _ptr(rates[0]),
             _ptr(rates[1]), _ptr(rates[2]), B, Cc, H, W, eng.dt, _stream())
        tape = [] if ctx.needs_input_grad[4] else None
        pred = eng.forward(xt, rates[2], model.training, tape, model._pop_drop_masks())
        nchunk = _lib.lib.vamd_l1_chunks(B * H * W)
        part = torch.empty(nchunk, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        call("vamd_l1_loss_fwd", _ptr(pred), _ptr(noises), _ptr(part), _ptr(loss), B, Cc, H, W, eng.dt, _stream())
        ctx.eng, ctx.tape, ctx.pred, ctx.noises, ctx.shape = eng, tape, pred, noises, (B, Cc, H, W)
        model._last_pred_nhwc = pred
        return loss

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.eng
        B, Cc, H, W = ctx.shape
        dpred = torch.empty_like(ctx.pred)
        g = gout.reshape(1).contiguous().float()
        call("vamd_l1_loss_bwd", _ptr(ctx.pred), _ptr(ctx.noises), _ptr(g), _ptr(dpred), B, Cc, H, W, eng.dt, _stream())
        eng.backward(ctx.tape, dpred)
        ctx.tape = ctx.pred = None
        return None, None, None, None, None


class DDPMModel(LightningModule):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        plan = make_plan(config)                       # raises KeyError for a missing key, like the reference
        self._plan = plan
        self.h_dims = config["h_dims"]
        self.patch_size = config["patch_size"]
        self.act = nn.LeakyReLU()
        self.attn_resolutions = config["attn_res"]
        self.down_index = config["down_index"]
        self.with_conv = config["with_conv"]
        self.dropout = config["dropout"]
        self.num_channels = len(self.h_dims)
        self.latent_dim = config["latent_dim"]
        self.sampling_period = config["sampling_period"]
        self.diffusion_steps = config["diffusion_steps"]
        self.grid = config["grid"]
        self.ch = self.h_dims[1]
        dt = kwargs.get("compute_dtype", config.get("compute_dtype", "bf16"))
        self.compute_dtype = dt if isinstance(dt, torch.dtype) else _DTYPES[dt]
        self.kernel_algo = int(kwargs.get("kernel_algo", config.get("kernel_algo", _lib.ALGO_AUTO)))

        # creation order = RNG order of the default initialisers (reference models/ddpm.py:156-204)
        self.linear_1 = _conv(self.ch, self.ch * 4, 1)
        self.linear_2 = _conv(self.ch * 4, self.ch * 4, 1)
        self.time_features = self.ch * 4
        sections = {"Downsampling": nn.ModuleList(), "Middle": nn.ModuleList(), "Upsampling": nn.ModuleList()}
        for spec in plan.layers:
            if spec.kind == "res":
                m = _ResParams(spec.cin, spec.cout, self.act, self.time_features, spec.dropout)
            elif spec.kind == "attn":
                m = _AttnParams(spec.cout)
            elif spec.kind == "down":
                m = _DownParams(spec.cout, self.with_conv)
            else:
                m = _UpParams(spec.cin, self.with_conv)
            sections[spec.section].append(m)
        self.Downsampling = sections["Downsampling"]
        self.Middle = sections["Middle"]
        self.reversed_h_dims = self.h_dims[::-1]
        self.reversed_down_index = self.down_index[::-1]
        self.Upsampling = sections["Upsampling"]
        self.End = nn.ModuleList([nn.Sequential(_conv(plan.end_channels, plan.end_channels, 1))])
        self.train_losses = []
        self.val_losses = []
        self._engine = None
        self._drop_masks = None
        self._last_pred_nhwc = None
        self.grad_reducer = None          # set by the data-parallel trainer

    # ------------------------------------------------------------------ engine plumbing
    def _ensure_engine(self) -> Engine:
        eng = self._engine
        if eng is None or not eng.owns(self) or eng.dtype != self.compute_dtype:
            eng = Engine(self, self._plan, self.compute_dtype, self.kernel_algo)
            object.__setattr__(self, "_engine", eng)
        return eng

    def set_dropout_masks(self, masks):
        """Testing hook: {'Middle.0': mask, 'Middle.2': mask} (NHWC; non-zero = keep, so both 0/1 masks
        and nn.Dropout's outputs pre-scaled by 1/(1-p) are accepted) used by the next training forward
        instead of fresh Bernoulli draws."""
        self._drop_masks = None if masks is None else {k: (v != 0).to(v.dtype) for k, v in masks.items()}

    def _pop_drop_masks(self):
        m, self._drop_masks = self._drop_masks, None
        return m

    def _anchor(self):
        """The parameter through which autograd reaches the engine's backward.  Without grad mode it is handed over
        detached: Function.forward then sees that no backward can follow and keeps no tape (validation, sampling)."""
        return self.linear_1.weight if torch.is_grad_enabled() else self.linear_1.weight.detach()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        if self._engine is not None:
            self._engine.dirty = True

    # ------------------------------------------------------------------ reference surface
    def forward(self, x, t=None):
        self.device = x.device                                    # reference models/ddpm.py:209
        if t is None:
            print("Test mode. It can't be trained.")
            t = torch.zeros((x.shape[0], 1, 1, 1), device=x.device)
        return _UNetFn.apply(self, x.contiguous().float(), t, self._anchor())

    def diffusion_schedule(self, diffusion_times):
        """(noise_rates, signal_rates) of the cosine schedule; reference models/ddpm.py:242-255."""
        t = torch.as_tensor(diffusion_times, dtype=torch.float32, device=self._anchor().device)
        flat = t.reshape(-1).contiguous()
        out = torch.empty((3, flat.numel()), dtype=torch.float32, device=flat.device)
        call("vamd_schedule", _ptr(flat), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), flat.numel(), _stream())
        return out[0].reshape(t.shape), out[1].reshape(t.shape)

    def denoise(self, x_t, noise_rates, signal_rates):
        pred_noises = self(x_t, noise_rates ** 2)
        pred_images = (x_t - noise_rates * pred_noises) / signal_rates
        return pred_noises, pred_images

    def p_loss(self, images, noises=None, diffusion_times=None):
        """L1 noise-prediction loss (reference models/ddpm.py:262-269).  `noises` / `diffusion_times`
        default to fresh draws from torch's generator in the reference's order (randn_like, then
        rand); passing them pins the step for parity tests."""
        images = images.contiguous().float()
        if noises is None:
            noises = torch.randn_like(images)
        if diffusion_times is None:
            diffusion_times = torch.rand((images.shape[0], 1, 1, 1), device=images.device)
        self.device = images.device
        return _PLossFn.apply(self, images, noises.contiguous().float(),
                              diffusion_times.reshape(-1).contiguous().float(), self._anchor())

    @torch.no_grad()
    def reverse_diffusion(self, initial_noise, diffusion_steps):
        """Deterministic sampler of reference models/ddpm.py:271-288; returns the last pred_images.
        Runs whatever rows it is given: callers shard `initial_noise` across ranks, no communication."""
        eng = self._ensure_engine()
        x = initial_noise.contiguous().float()
        B, Cc, H, W = x.shape
        dev = x.device
        T = int(diffusion_steps)
        step = 1.0 / T
        # diffusion times exactly as the reference forms them in float32: ones - step*step_size, then - step_size
        t_now = np.array([np.float32(1.0) - np.float32(k * step) for k in range(T)], dtype=np.float32)
        t_next = (t_now - np.float32(step)).astype(np.float32)
        tt = torch.from_numpy(np.concatenate([t_now, t_next])).to(dev)
        rates = torch.empty((3, 2 * T), dtype=torch.float32, device=dev)
        call("vamd_schedule", _ptr(tt), _ptr(rates[0]), _ptr(rates[1]), _ptr(rates[2]), 2 * T, _stream())
        # per-step, per-sample rate rows: [T][5][B] = n, s, n^2, n_next, s_next
        table = torch.stack([rates[0, :T], rates[1, :T], rates[2, :T], rates[0, T:], rates[1, T:]], dim=1)
        table = table[:, :, None].expand(T, 5, B).contiguous()
        cur = torch.empty((B, H, W, Cc), dtype=torch.float32, device=dev)
        xin = torch.empty((B, H, W, Cc), dtype=eng.dtype, device=dev)
        call("vamd_nchw_to_nhwc", _ptr(x), _ptr(xin), _ptr(cur), B, Cc, H, W, eng.dt, _stream())
        pred_images = torch.empty((B, Cc, H, W), dtype=torch.float32, device=dev)
        was_training = self.training
        for k in range(T):
            r = table[k]
            eps_hat = eng.forward(xin, r[2], self.training, None)
            last = k == T - 1
            call("vamd_denoise_step", _ptr(cur), _ptr(eps_hat), _ptr(r[0]), _ptr(r[1]), _ptr(r[3]), _ptr(r[4]),
                 _ptr(xin), _ptr(pred_images) if last else None, B, Cc, H, W, eng.dt, _stream())
        assert self.training == was_training
        return pred_images

    def generate(self, num_images, diffusion_steps, initial_noise=None):
        if initial_noise is None:
            initial_noise = torch.randn((num_images, 3, self.patch_size, self.patch_size),
                                        device=self._anchor().device)
        return self.reverse_diffusion(initial_noise, diffusion_steps)

    # ------------------------------------------------------------------ loops (reference :301-368)
    def save_loss(self, phase, loss):
        save_path = os.path.join(self.logger.log_dir, f"{phase}_losses.npy")
        if self.trainer.is_global_zero:
            try:
                losses = np.append(np.load(save_path), loss)
            except FileNotFoundError:
                losses = np.array([loss])
            np.save(save_path, losses)

    def training_step(self, batch, batch_idx):
        x_0, y = batch
        loss = self.p_loss(x_0)
        self.log("TL", loss, on_step=True, on_epoch=True, prog_bar=True, sync_dist=True)
        # the reference calls loss.item() here (models/ddpm.py:316), a host sync per step; the
        # value is kept on the device and read once per epoch instead
        self.train_losses.append(loss.detach())
        return loss

    def on_train_epoch_end(self):
        avg_loss = float(np.mean([float(l) for l in self.train_losses])) if self.train_losses else float("nan")
        self.train_losses = []
        self.save_loss("train", avg_loss)

    def validation_step(self, batch, batch_idx):
        x_0, y = batch
        loss = self.p_loss(x_0)
        self.log("VL", loss, on_step=False, on_epoch=True, prog_bar=True, sync_dist=True)
        self.val_losses.append(loss.detach())

    def on_validation_end(self) -> None:
        if self.current_epoch % self.sampling_period == 0:
            self.sample_images()

    def test_step(self, batch, batch_idx):
        x_0, y = batch
        loss = self.p_loss(x_0)
        self.log("TeL", loss, on_step=False, on_epoch=True, sync_dist=True)

    def configure_optimizers(self):
        optimizer = FusedAdamW(self, lr=self.config["learning_rate"], weight_decay=self.config["weight_decay"])
        # reference quirks kept (models/ddpm.py:347-352): a missing key returns the bare optimiser,
        # an explicit None falls through and returns None
        try:
            if self.config["scheduler_gamma"] is not None:
                scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=self.config["scheduler_gamma"])
                return [optimizer], [scheduler]
        except Exception:
            return optimizer

    def sample_images(self):
        try:
            samples = self.generate(self.grid ** 2, self.diffusion_steps)
            samples_dir = os.path.join(self.logger.log_dir, "Samples")
            os.makedirs(samples_dir, exist_ok=True)
            save_image_grid(samples.cpu(), os.path.join(samples_dir, f"{self.logger.name}_Epoch_{self.current_epoch}.png"),
                            normalize=True, nrow=self.grid)
        except Warning:
            pass

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, config, *args, **kwargs):
        from safetensors.torch import load_file
        model = cls(config, *args, **kwargs)
        model.load_state_dict(load_file(checkpoint_path), strict=False)
        return model
