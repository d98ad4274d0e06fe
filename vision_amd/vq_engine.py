"""Execution engine of the VQ-VAE-UNet first stage (reference models/vqvaeunet.py:149-233): the UNet engine's layer
objects, arenas and weight re-packing, with the first stage's own wiring:

* ResNetBlocks without a time embedding whose 1x1 shortcut is a channel-wise nn.Linear (key `layers.0.linear`); the
  channel-attention block of the latent-diffusion file where attn_res asks for one;
* EVERY encoder module's output is a skip, added to the input of the decoder module at the mirrored position
  (`h += skip[index]`, models/vqvaeunet.py:228-231) -- here by the epilogue of the kernel producing that input;
* the codebook lookup between the two halves, in fp32 on the encoder's NHWC output (its rows are the latents, so no
  permute is needed), with the straight-through output, the loss and both gradients from csrc/vq.hip;
* a channel-wise Linear head; its sigmoid is fused with the reconstruction loss by the caller.
"""
from typing import Dict, List, Optional

import contextlib

import torch

from ._lib import BF16, F32, call, lib
from .engine import FUSE_EVAL_NORM, Engine, _ChanAttnBlock, _Conv, _ResBlock, _Resample, _ptr, _stream


class VQUNetEngine(Engine):
    shortcut_key = ".layers.0.linear"      # Custumnn.Linear wraps an nn.Linear (reference models/Custumnn.py:5-8)

    def __init__(self, module, plan, dtype, beta, algo=0):
        self.beta = float(beta)
        super().__init__(module, plan, dtype, algo)

    def _init_layers(self, plan):
        self.layers = []
        for spec in plan.layers:
            if spec.kind == "res":
                self.layers.append(_ResBlock(self, spec, 0))
            elif spec.kind == "attn":
                self.layers.append(_ChanAttnBlock(self, spec))     # models/vqvaeunet.py:83-100, the block of models/ldm.py
            else:
                self.layers.append(_Resample(self, spec))
        self.n_enc = len(plan.section("Downsampling"))
        n_dec = len(plan.section("Upsampling"))
        head = f"Upsampling.{n_dec}"
        self.end = _Conv(self, head + ".0.linear", plan.end_channels, plan.end_channels, 1)
        self.codebook = self.param("vq_layer.embedding.weight")
        self.g_codebook = self.grad("vq_layer.embedding.weight")
        self._lows = {spec.prefix: self._layer_low(spec.prefix) for spec in plan.layers}
        self._lows["End"] = self._layer_low(head)
        self.last_indices = None

    def _project_temb(self, temb, B):       # no time embedding in this model
        self.tproj_for, self.tproj_src = {}, None

    _DT = {torch.float32: F32, torch.bfloat16: BF16}

    def _add_cast(self, a, b, dtype):
        """a (+ b) in `dtype`, summed in fp32 (vamd_add_cast): the fp32 <-> compute-dtype steps around the lookup."""
        a = a.contiguous()
        out = torch.empty(a.shape, dtype=dtype, device=a.device)
        if b is not None:
            b = b.contiguous()
            assert b.shape == a.shape
        call("vamd_add_cast", _ptr(a), self._DT[a.dtype], _ptr(b), self._DT[b.dtype] if b is not None else 0, _ptr(out),
             self._DT[dtype], a.numel(), _stream())
        return out

    # ---- forward / backward on NHWC tensors
    def _begin(self, training):
        self.refresh()
        self.stats_for.clear()
        self.act_for.clear()
        if training:
            self.coef_epoch += 1

    def _run(self, first, last, h, skips, training, tape):
        """Layers first..last-1.  Encoder outputs are appended to `skips`; the input of decoder module j is (output of
        module j-1) + skips[ne-1-j] (reference models/vqvaeunet.py:228-231): for j >= 1 the producer's epilogue adds it."""
        ne, n = self.n_enc, len(self.layers)
        for i in range(first, last):
            layer = self.layers[i]
            j = i + 1 - ne
            res2 = skips[ne - 1 - j] if 1 <= j < ne else None
            if res2 is not None and tuple(res2.shape) != self._out_shape(layer, h):
                # the reference fails here too (`h += skip[index]` on mismatched shapes): an AttnBlock at a level whose
                # block changes the channel count pairs encoder and decoder outputs of different widths
                raise RuntimeError(f"skip of shape {tuple(res2.shape)} meets the output {self._out_shape(layer, h)} of "
                                   f"{layer.spec.prefix}: encoder and decoder do not mirror each other in this configuration")
            nxt = self.layers[i + 1] if i + 1 < n and i + 1 != ne else None       # nothing is fused across the lookup
            want = training and isinstance(nxt, _ResBlock)
            nn_ = nxt.bn1 if (FUSE_EVAL_NORM and not training and tape is None and isinstance(nxt, _ResBlock)) else None
            h = layer.fwd(h, None, training, res2, tape, None, want_stats=want, next_norm=nn_)
            if self.debug_taps is not None:
                self.debug_taps[layer.spec.prefix] = (h, res2)
            if i < ne:
                skips.append(h)
        return h

    @staticmethod
    def _out_shape(layer, h):
        B, H, W, _ = h.shape
        if layer.spec.kind == "down":
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        elif layer.spec.kind == "up":
            H, W = 2 * H, 2 * W
        return (B, H, W, layer.spec.cout)

    def encode(self, x, training=False, tape=None):
        """reference models/vqvaeunet.py:215-222: (encoder output, every encoder module's output), NHWC."""
        self._begin(training)
        skips: List[torch.Tensor] = []
        return self._run(0, self.n_enc, x, skips, training, tape), skips

    def decode(self, h, skips, training=False, tape=None):
        """reference models/vqvaeunet.py:224-233 up to the head's Linear (the sigmoid belongs to the caller's kernel).
        `h` may be fp32 (the codebook lookup's output) or the compute dtype; the first skip is added here."""
        if len(skips) != self.n_enc:
            raise ValueError(f"decode() needs the {self.n_enc} encoder outputs, got {len(skips)}")
        if tape is None:
            self._begin(training)          # a call of its own (inference): weights may have changed since the last encode
        last = skips[-1]
        if h.shape != last.shape:
            raise ValueError(f"latents {tuple(h.shape)} do not match the encoder output {tuple(last.shape)}")
        h = self._add_cast(h, last, self.dtype)
        h = self._run(self.n_enc, len(self.layers), h, skips, training, tape)
        out = self.end.fwd(h)
        if tape is not None:
            tape.append(("end", (h,)))
        return out

    def forward(self, x, training: bool, tape: Optional[list]):
        """x: [B,H,W,3] compute dtype.  Returns (head output before the sigmoid, NHWC; vq_loss, a 0-d fp32 tensor)."""
        h, skips = self.encode(x, training, tape)
        q, vq_loss = self.quantize(h, tape)
        return self.decode(q, skips, training, tape), vq_loss

    def quantize(self, h, tape=None):
        """reference models/vqvaeunet.py:115-146 on the rows of an NHWC tensor: (straight-through output fp32, vq_loss)."""
        B, Hh, Wh, D = h.shape
        K = self.codebook.shape[0]
        if self.codebook.shape[1] != D:
            raise ValueError(f"latents have {D} channels, the codebook {self.codebook.shape[1]} features")
        M = B * Hh * Wh
        z = h if h.dtype == torch.float32 else self._add_cast(h, None, torch.float32)
        dev = h.device
        ws = torch.empty(lib.vamd_vq_workspace(M), dtype=torch.uint8, device=dev)
        idx = torch.empty(M, dtype=torch.int64, device=dev)
        out = torch.empty_like(z)
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        call("vamd_vq_forward", _ptr(z), _ptr(self.codebook), M, D, K, Hh * Wh, 0, self.beta, _ptr(idx), _ptr(out),
             _ptr(loss), _ptr(ws), ws.numel(), _stream())
        self.last_indices = idx
        if tape is not None:
            tape.append(("vq", (z, idx)))
        return out, loss[0]

    def backward(self, tape: list, dout, g_vq):
        """dout: gradient of the head output (NHWC); g_vq: gradient of vq_loss (one device float, or None).
        Overwrites every parameter gradient in the flat arena."""
        tag, (h_end,) = tape.pop()
        assert tag == "end"
        red = self.reducer
        if red is not None:
            red.begin()
        self._side_begin(dout.device)                 # weight gradients on the second stream, as in Engine.backward
        self.end.wgrad(dout, h_end)
        g = self.end.dgrad(dout, h_end.shape)
        live = red is not None and getattr(red, "_on", True)
        if red is not None:
            with self.side_section(torch_ops=True) if live else contextlib.nullcontext():
                red.layer_done(self._lows["End"])
        ne, n = self.n_enc, len(self.layers)
        dskip: Dict[int, torch.Tensor] = {}
        for i in range(n - 1, -1, -1):
            layer, saved = tape.pop()
            assert layer is self.layers[i]
            # the skip gradient of encoder output i-1 is folded into this layer's input gradient
            extra = dskip.pop(i - 1, None) if 1 <= i < ne else None
            g = layer.bwd(saved, g, extra, None, need_dx=(i > 0))
            j = i - ne
            if 0 <= j < ne:
                dskip[ne - 1 - j] = g                   # the decoder input was a sum: its gradient reaches the skip too
            if red is not None and self._lows[layer.spec.prefix] is not None:
                with self.side_section(torch_ops=True) if live else contextlib.nullcontext():
                    red.layer_done(self._lows[layer.spec.prefix])
            if i == ne:
                g = self._quantize_bwd(tape, g, dskip.pop(ne - 1), g_vq)
        if red is not None:
            with self.side_section(torch_ops=True) if live else contextlib.nullcontext():
                red.finish()
        self._side_join()
        self.attach_grads()

    def _quantize_bwd(self, tape, g_out, g_skip, g_vq):
        tag, (z, idx) = tape.pop()
        assert tag == "vq"
        M, D = idx.numel(), z.shape[-1]
        K = self.codebook.shape[0]
        g32 = g_out if g_out.dtype == torch.float32 else self._add_cast(g_out, None, torch.float32)
        dz = torch.empty_like(z)
        call("vamd_vq_backward", _ptr(z), _ptr(self.codebook), _ptr(idx), _ptr(g32), _ptr(g_vq), M, D, K,
             z.shape[1] * z.shape[2], 0, self.beta, _ptr(dz), _ptr(self.g_codebook), _stream())
        return self._add_cast(dz, g_skip, self.dtype)           # + the gradient that reached the same tensor as a skip
