"""MI355X-native `VQVAEUNETModel`: the LightningModule surface of the reference's first-stage autoencoder
(reference models/vqvaeunet.py:149-327: constructor, attributes, method names, state_dict keys, YAML schema) on the
HIP engine.

As with the DDPM model, parameters live in torch containers with the reference's module tree -- so `state_dict()`
keys and shapes, default initialisation and its RNG order are the reference's and safetensors checkpoints interchange
both ways -- and the containers are never called: libvision_amd.so does the arithmetic (vision_amd.vq_engine).
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from ._lib import call
from .ddpm import _DTYPES, _conv
from .engine import _ptr, _stream
from .lightning_shim import LightningModule
from .optim import FusedAdamL2
from .plan import make_vq_plan
from .utils import save_image_grid
from .vq import VQBlock
from .vq_engine import VQUNetEngine


class _ChannelLinear(nn.Module):
    """Key layout of the reference's Custumnn.Linear (models/Custumnn.py:5-8): an nn.Linear applied over the channel axis."""

    def __init__(self, cin, cout):
        super().__init__()
        self.linear = nn.Linear(cin, cout)


class _ResParams(nn.Module):
    """Parameter holder with the key layout and creation order of the reference block (models/vqvaeunet.py:44-68)."""

    def __init__(self, cin, cout, act, dropout=0.0):
        super().__init__()
        self.act = act
        self.block1 = nn.ModuleList([nn.Sequential(nn.BatchNorm2d(cin), act, _conv(cin, cout, 3, padding=1))])
        self.block2 = nn.ModuleList([nn.Sequential(nn.BatchNorm2d(cout), act, nn.Dropout(dropout),
                                                   _conv(cout, cout, 3, padding=1))])
        self.layers = nn.ModuleList([_ChannelLinear(cin, cout)]) if cin != cout else None


class _AttnParams(nn.Module):
    """reference models/vqvaeunet.py:83-88."""

    def __init__(self, c):
        super().__init__()
        self.linear1 = _ChannelLinear(c, c)
        self.linear2 = _ChannelLinear(c, c)
        self.norm = nn.BatchNorm2d(c)


class _DownParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layers = nn.ModuleList([_conv(c, c, 3, stride=2, padding=1)])


class _UpParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layers = nn.ModuleList([nn.Upsample(scale_factor=2, mode="nearest"), _conv(c, c, 3, padding=1)])


class _FirstStageFn(torch.autograd.Function):
    """encode -> codebook lookup -> decode -> sigmoid (-> reconstruction loss) as one autograd node."""

    @staticmethod
    def forward(ctx, model, x, with_loss, anchor):
        if ctx.needs_input_grad[1]:
            raise RuntimeError("vision_amd computes parameter gradients only: the images must not require grad")
        eng = model._ensure_engine()
        B, Cc, H, W = x.shape
        dev = x.device
        xn = torch.empty((B, H, W, Cc), dtype=eng.dtype, device=dev)
        call("vamd_nchw_to_nhwc", _ptr(x), _ptr(xn), None, B, Cc, H, W, eng.dt, _stream())
        tape = [] if ctx.needs_input_grad[3] else None        # grad mode is off inside forward()
        h, vq_loss = eng.forward(xn, model.training, tape)
        if h.shape[3] != Cc:
            raise RuntimeError(f"the decoder ends in {h.shape[3]} channels, the images have {Cc}")
        xrec = torch.empty((B, Cc, H, W), dtype=torch.float32, device=dev)
        part = rec = None
        if with_loss:
            part = torch.empty(_lib.lib.vamd_l1_chunks(B * H * W), dtype=torch.float32, device=dev)
            rec = torch.empty((), dtype=torch.float32, device=dev)
        call("vamd_sigmoid_mse_fwd", _ptr(h), _ptr(x) if with_loss else None, _ptr(xrec), _ptr(part), _ptr(rec),
             B, Cc, H, W, eng.dt, _stream())
        # the output and the input go through save_for_backward: kept as plain attributes, output -> grad_fn -> ctx ->
        # output would be a cycle the collector cannot see, and a grad-mode forward that is never backpropagated would
        # keep its whole tape on the device
        ctx.save_for_backward(xrec, x)
        ctx.eng, ctx.tape, ctx.with_loss, ctx.shape = eng, tape, with_loss, (B, Cc, H, W)
        ctx.set_materialize_grads(False)
        if with_loss:
            return xrec, vq_loss, rec
        return xrec, vq_loss

    @staticmethod
    def backward(ctx, g_xrec, g_vq, g_rec=None):
        eng = ctx.eng
        B, Cc, H, W = ctx.shape
        xrec, x = ctx.saved_tensors
        dev = xrec.device
        dh = torch.empty((B, H, W, Cc), dtype=eng.dtype, device=dev)
        f32 = lambda t: None if t is None else t.contiguous().float()
        g_xrec, g_rec = f32(g_xrec), (None if g_rec is None else f32(g_rec).reshape(1))
        call("vamd_sigmoid_mse_bwd", _ptr(xrec), _ptr(x) if (g_rec is not None and ctx.with_loss) else None, _ptr(g_rec), _ptr(g_xrec),
             _ptr(dh), B, Cc, H, W, eng.dt, _stream())
        eng.backward(ctx.tape, dh, None if g_vq is None else f32(g_vq).reshape(1))
        ctx.tape = None
        return None, None, None, None


class VQVAEUNETModel(LightningModule):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        plan = make_vq_plan(config)                    # raises KeyError for a missing key, like the reference
        self._plan = plan
        self.h_dims = config["h_dims"]
        self.patch_size = config["patch_size"]
        self.num_embed = config["num_embed"]
        self.embed_dim = config["embed_dim"]
        self.sampling_period = config["sampling_period"]
        self.attn_resolutions = config["attn_res"]
        self.down_index = config["down_index"]
        self.act = nn.LeakyReLU()
        self.with_conv = True
        dt = kwargs.get("compute_dtype", config.get("compute_dtype", "bf16"))
        self.compute_dtype = dt if isinstance(dt, torch.dtype) else _DTYPES[dt]
        self.kernel_algo = int(kwargs.get("kernel_algo", config.get("kernel_algo", _lib.ALGO_AUTO)))

        # creation order = RNG order of the default initialisers (reference models/vqvaeunet.py:163-203)
        self.vq_layer = VQBlock(self.num_embed, self.embed_dim)
        sections = {"Downsampling": nn.ModuleList(), "Upsampling": nn.ModuleList()}
        for spec in plan.layers:
            if spec.kind == "res":
                m = _ResParams(spec.cin, spec.cout, self.act)
            elif spec.kind == "attn":
                m = _AttnParams(spec.cout)
            elif spec.kind == "down":
                m = _DownParams(spec.cout)
            else:
                m = _UpParams(spec.cin)
            sections[spec.section].append(m)
        self.Downsampling = sections["Downsampling"]
        self.reversed_h_dims = self.h_dims[::-1]
        self.reversed_down_index = self.down_index[::-1]
        self.Upsampling = sections["Upsampling"]
        self.Upsampling.append(nn.Sequential(_ChannelLinear(plan.end_channels, plan.end_channels), nn.Sigmoid()))
        self._engine = None
        self.grad_reducer = None          # set by the data-parallel trainer
        self.curr_device = None

    # ------------------------------------------------------------------ engine plumbing
    def _ensure_engine(self) -> VQUNetEngine:
        eng = self._engine
        if eng is None or not eng.owns(self) or eng.dtype != self.compute_dtype:
            eng = VQUNetEngine(self, self._plan, self.compute_dtype, self.vq_layer.beta, self.kernel_algo)
            object.__setattr__(self, "_engine", eng)
        return eng

    def _anchor(self):
        """The parameter through which autograd reaches the engine's backward; detached without grad mode, so that no
        tape is kept for forwards no backward can follow."""
        w = self.vq_layer.embedding.weight
        return w if torch.is_grad_enabled() else w.detach()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        if self._engine is not None:
            self._engine.dirty = True

    def _nhwc(self, x):
        eng = self._ensure_engine()
        B, Cc, H, W = x.shape
        xn = torch.empty((B, H, W, Cc), dtype=eng.dtype, device=x.device)
        call("vamd_nchw_to_nhwc", _ptr(x.contiguous().float()), _ptr(xn), None, B, Cc, H, W, eng.dt, _stream())
        return xn

    def _nchw(self, h):
        eng = self._ensure_engine()
        B, H, W, Cc = h.shape
        h = h if h.dtype == eng.dtype else h.to(eng.dtype)
        out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=h.device)
        call("vamd_nhwc_to_nchw", _ptr(h.contiguous()), _ptr(out), B, Cc, H, W, eng.dt, _stream())
        return out

    # ------------------------------------------------------------------ reference surface
    def forward(self, x):
        """[x_rec, vq_loss] (reference models/vqvaeunet.py:205-213); x_rec is NCHW fp32 in (0, 1)."""
        xrec, vq_loss = _FirstStageFn.apply(self, x.contiguous().float(), False, self._anchor())
        return [xrec, vq_loss]

    @torch.no_grad()
    def encode(self, x):
        """(h, skip) of reference models/vqvaeunet.py:215-222 as NCHW fp32 tensors; inference only (the training
        path is forward() / training_step(), one autograd node)."""
        eng = self._ensure_engine()
        h, skips = eng.encode(self._nhwc(x), self.training)
        return self._nchw(h), [self._nchw(s) for s in skips]

    @torch.no_grad()
    def encode_latents(self, x):
        """The encoder output alone (what a latent-diffusion model trains on), NCHW fp32: encode() without the
        layout conversion of every skip."""
        eng = self._ensure_engine()
        h, _ = eng.encode(self._nhwc(x), self.training)
        return self._nchw(h)

    @torch.no_grad()
    def quantize(self, h):
        """The codebook lookup on NCHW latents: (quantised latents NCHW fp32, vq_loss, codes [B, H, W] int64)."""
        eng = self._ensure_engine()
        z = h.contiguous().float().permute(0, 2, 3, 1).contiguous()
        q, vq_loss = eng.quantize(z)
        return q.permute(0, 3, 1, 2).contiguous(), vq_loss, eng.last_indices.view(z.shape[:3])

    @torch.no_grad()
    def decode(self, x, skip):
        """reference models/vqvaeunet.py:224-233 on NCHW tensors; `skip` in the order encode() returned it (the
        reference reverses the caller's list in place, :225; this method leaves it alone).  Inference only."""
        eng = self._ensure_engine()
        h = eng.decode(x.contiguous().float().permute(0, 2, 3, 1).contiguous(), [self._nhwc(s) for s in skip], self.training)
        B, H, W, Cc = h.shape
        xrec = torch.empty((B, Cc, H, W), dtype=torch.float32, device=h.device)
        call("vamd_sigmoid_mse_fwd", _ptr(h), None, _ptr(xrec), None, None, B, Cc, H, W, eng.dt, _stream())
        return xrec

    def loss_function(self, x, x_rec, vq_loss):
        return F.mse_loss(x_rec, x) + vq_loss                 # reference models/vqvaeunet.py:235-239

    def _step_loss(self, batch):
        x, _ = batch
        self.curr_device = x.device
        _, vq_loss, rec = _FirstStageFn.apply(self, x.contiguous().float(), True, self._anchor())
        return rec + vq_loss

    def training_step(self, batch, batch_idx):
        loss = self._step_loss(batch)
        self.log("TL", loss, on_step=True, on_epoch=True, prog_bar=True, sync_dist=True)
        return loss

    def validation_step(self, batch, batch_idx):
        loss = self._step_loss(batch)
        self.log("VL", loss, on_step=False, on_epoch=True, prog_bar=True, sync_dist=True)

    def test_step(self, batch, batch_idx):
        loss = self._step_loss(batch)
        self.log("TeL", loss, on_step=False, on_epoch=True, sync_dist=True)

    def configure_optimizers(self):
        optimizer = FusedAdamL2(self, lr=self.config["learning_rate"], weight_decay=self.config["weight_decay"])
        # reference quirks kept (models/vqvaeunet.py:271-276): a missing key returns the bare optimiser, an explicit
        # None falls through and returns None
        try:
            if self.config["scheduler_gamma"] is not None:
                scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=self.config["scheduler_gamma"])
                return [optimizer], [scheduler]
        except Exception:
            return optimizer

    def on_validation_epoch_end(self):
        if self.current_epoch % self.sampling_period == 0:
            self.sample_images()

    @torch.no_grad()
    def sample_images(self):
        """Reconstructions of the first test batch (reference models/vqvaeunet.py:293-306)."""
        test_input, test_label = next(iter(self.trainer.datamodule.test_dataloader()))
        test_input = test_input.to(self.curr_device)
        recons = self(test_input)[0]
        recons_dir = os.path.join(self.logger.log_dir, "Reconstructions")
        os.makedirs(recons_dir, exist_ok=True)
        save_image_grid(recons.cpu(), os.path.join(recons_dir, f"recons_{self.logger.name}_Epoch_{self.current_epoch}.png"),
                        normalize=True, nrow=4)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, config, *args, **kwargs):
        from safetensors.torch import load_file
        model = cls(config, *args, **kwargs)
        model.load_state_dict(load_file(checkpoint_path), strict=False)
        return model
