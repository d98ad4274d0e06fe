"""Execution engine: runs the DDPM UNet forward and its hand-derived backward as a fixed sequence
of libvision_amd.so kernel launches on the current HIP stream.

There is no tracing and no per-op autograd: `forward` records, per layer, exactly the tensors its
backward needs, and `backward` walks the layers in reverse, fusing every gradient merge (residual
shortcuts, UNet skip connections) into the kernel that produces the gradient.  torch is used for
device memory (torch.empty on the caching allocator), streams and parameters only.

Layouts: activations NHWC in the compute dtype (bf16 or fp32); parameters stay fp32 in torch's
[Cout][Cin][kh][kw] layout inside one flat arena (so state_dict()/safetensors interchange with
the reference), with packed compute-dtype copies refreshed after every optimiser step.
"""
import contextlib
import ctypes as C
import math
import os
import threading
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import ConvDesc, LinearBwdItem, LinearItem, RepackItem, call, lib
from .plan import LayerSpec, UNetPlan

LEAKY_SLOPE = 0.01          # nn.LeakyReLU() default (reference models/ddpm.py:144)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
GN_EPS, GN_GROUPS = 1e-6, 32


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


class _StreamOverride(threading.local):
    raw = None               # raw handle of the second stream while one of Engine.side_section's blocks runs in this thread


_override = _StreamOverride()


def _stream():
    """The HIP stream the library's launches go to: the current stream's handle, or the engine's second stream inside a
    side section.  (torch.cuda.current_stream().cuda_stream builds a Stream object through several layers of device
    look-ups, ~8 us a call; a step makes ~400 of them.)"""
    raw = _override.raw
    return raw if raw is not None else _raw_stream(_cur_device())


# a BatchNorm + LeakyReLU in front of a convolution applied inside the convolution's forward and weight-gradient kernels
# (shapes for which both have a kernel that stages the input patch in LDS): the normalised tensor is never stored.
# Opt-in: bit-identical, saves one write and two reads of the tensor, but the two kernels are bound by their LDS traffic
# and the extra pass over the staged patch costs them more than the separate pass takes (64 x 224 x 224 x 64: separate
# pass 160 us; forward 394 -> 457 us, weight gradient 238 -> 300 us; smaller layers lose 20 - 70 us)
FUSE_TRAIN_NORM = os.environ.get("VAMD_PRENORM", "0") != "0"


class _ShapePlan:
    """What a convolution's launches need to know about one input shape, asked of the library once."""
    __slots__ = ("d", "ref", "up4_fwd", "parts", "parts_up4", "up4_dgrad", "dgrad_ws", "wgrad_ws", "act_ok", "mfma", "prenorm")

    def __init__(self, d, has_up4):
        self.d, self.ref = d, C.byref(d)
        self.up4_fwd = bool(has_up4 and lib.vamd_conv2d_fwd_up4_supported(self.ref))
        self.up4_dgrad = bool(has_up4 and lib.vamd_conv2d_dgrad_up4_supported(self.ref))
        self.parts = lib.vamd_conv2d_stats_parts(self.ref)
        self.parts_up4 = lib.vamd_conv2d_fwd_up4_stats_parts(self.ref) if self.up4_fwd else 0
        self.dgrad_ws = lib.vamd_conv2d_dgrad_workspace(self.ref)
        self.wgrad_ws = lib.vamd_conv2d_wgrad_workspace(self.ref)
        self.act_ok = bool(lib.vamd_conv2d_fwd_act_supported(self.ref))
        self.prenorm = bool(FUSE_TRAIN_NORM and lib.vamd_conv2d_prenorm_supported(self.ref))
        self.mfma = [bool(lib.vamd_conv2d_uses_mfma(self.ref, which)) for which in range(3)]


class _Conv:
    """One nn.Conv2d of the reference: fp32 master weight/bias views, packed copies, launches."""

    def __init__(self, eng: "Engine", name: str, cin: int, cout: int, k: int, stride=1, up=False):
        self.eng, self.name = eng, name
        self.cin, self.cout, self.k, self.stride, self.up = cin, cout, k, stride, up
        self.pad = 1 if k == 3 else 0
        self.w, self.b = eng.param(name + ".weight"), eng.param(name + ".bias")
        self.gw, self.gb = eng.grad(name + ".weight"), eng.grad(name + ".bias")
        self.wf = torch.empty((cout, k, k, cin), dtype=eng.dtype, device=eng.device)
        self.wd = torch.empty((cin, k, k, cout), dtype=eng.dtype, device=eng.device)
        # upsampling convs: combined 4x4 stride-2 filter of the input gradient (filled by repack_up4)
        self.wd4 = torch.empty((cin, 4, 4, cout), dtype=eng.dtype, device=eng.device) if up and k == 3 else None
        # ... and the four 2x2 class filters of their forward (one per output parity)
        self.wf4 = torch.empty((4, cout, 2, 2, cin), dtype=eng.dtype, device=eng.device) if up and k == 3 else None
        self._descs: Dict[tuple, ConvDesc] = {}
        self._plans: Dict[tuple, _ShapePlan] = {}

    def repack(self):
        call("vamd_repack_weight", _ptr(self.w), _ptr(self.wf), _ptr(self.wd), self.cout, self.cin,
             self.k, self.eng.dt, _stream())

    def repack_up4(self):
        if self.wd4 is not None:
            call("vamd_repack_weight_up4", _ptr(self.w), _ptr(self.wd4), self.cout, self.cin, self.eng.dt, _stream())
            call("vamd_repack_weight_up4_fwd", _ptr(self.w), _ptr(self.wf4), self.cout, self.cin, self.eng.dt, _stream())

    def desc(self, B, H, W) -> ConvDesc:
        key = (B, H, W)
        d = self._descs.get(key)
        if d is None:
            d = ConvDesc(B, H, W, self.cin, self.cout, self.k, self.stride, self.pad, int(self.up),
                         self.eng.dt, self.eng.algo)
            if self.eng.scratch is not None:      # lets the wide kernels split tiles between workgroups
                # (a convolution that may run beside another one on a stream of its own has its own buffer)
                buf = self.eng.scratch_aux if getattr(self, "own_scratch", False) else self.eng.scratch
                d.scratch, d.scratch_bytes = buf.data_ptr(), buf.numel()
            self._descs[key] = d
        return d

    def plan(self, B, H, W) -> _ShapePlan:
        key = (B, H, W)
        p = self._plans.get(key)
        if p is None:
            p = self._plans[key] = _ShapePlan(self.desc(B, H, W), self.wf4 is not None)
        return p

    def out_hw(self, H, W):
        if self.up:
            return 2 * H, 2 * W
        if self.stride == 2:
            return (H - 1) // 2 + 1, (W - 1) // 2 + 1
        return H, W

    def _timed(self, which, B, H, W):
        """Benchmark hook: when eng.profile is a list, bracket MFMA launches with HIP events."""
        prof = self.eng.profile
        if prof is None or not self.plan(B, H, W).mfma[which]:
            return None
        Ho, Wo = self.out_hw(H, W)
        flops = 2.0 * B * Ho * Wo * self.cin * self.cout * self.k * self.k
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        prof.append(("tn" if which == 2 else "nt", flops, e0, e1,
                     (("fwd", "dgrad", "wgrad")[which], self.cin, self.cout, self.k, self.stride, int(self.up), B, H, W)))
        return e1

    def fwd(self, x, temb=None, res=None, res2=None, bias=True, want_stats=False, pre=None):
        """y = conv(x) + bias (+ temb + res + res2).  With want_stats, the kernel's epilogue also
        leaves the per-channel sums of y and y*y (what a train-mode BatchNorm over y needs) in
        eng.stats_for[y.data_ptr()], if the kernel serving this shape can produce them.
        pre = (scale, shift, slope): the convolution runs on leaky(x * scale + shift) rounded to the compute dtype, formed
        inside the kernel (only where plan(...).prenorm says so)."""
        B, H, W, _ = x.shape
        Ho, Wo = self.out_hw(H, W)
        e = self.eng
        y = torch.empty((B, Ho, Wo, self.cout), dtype=e.dtype, device=x.device)
        sp = self.plan(B, H, W)
        up4 = sp.up4_fwd
        nparts = (sp.parts_up4 if up4 else sp.parts) if want_stats else 0
        ev = self._timed(0, B, H, W) if e.profile is not None else None
        if pre is not None:
            part = torch.empty((nparts, self.cout, 2), dtype=torch.float32, device=x.device) if nparts > 0 else None
            call("vamd_conv2d_fwd_prenorm", sp.ref, _ptr(x), _ptr(pre[0]), _ptr(pre[1]), pre[2], _ptr(self.wf),
                 _ptr(self.b) if bias else None, _ptr(temb), _ptr(res), _ptr(res2), _ptr(y), _ptr(part), _stream())
        elif up4:
            part = torch.empty((nparts, self.cout, 2), dtype=torch.float32, device=x.device) if nparts > 0 else None
            call("vamd_conv2d_fwd_up4", sp.ref, _ptr(x), _ptr(self.wf4), _ptr(self.b) if bias else None,
                 _ptr(temb), _ptr(res), _ptr(res2), _ptr(y), _ptr(part), _stream())
        elif nparts > 0:
            part = torch.empty((nparts, self.cout, 2), dtype=torch.float32, device=x.device)
            call("vamd_conv2d_fwd_stats", sp.ref, _ptr(x), _ptr(self.wf), _ptr(self.b) if bias else None,
                 _ptr(temb), _ptr(res), _ptr(res2), _ptr(y), _ptr(part), _stream())
        else:
            call("vamd_conv2d_fwd", sp.ref, _ptr(x), _ptr(self.wf),
                 _ptr(self.b) if bias else None, _ptr(temb), _ptr(res), _ptr(res2), _ptr(y), _stream())
        if ev is not None:
            ev.record()
        if nparts > 0:
            if nparts > 512:                      # tens of thousands of row tiles at full resolution
                folded = torch.empty((128, self.cout, 2), dtype=torch.float32, device=x.device)
                call("vamd_fold_partials", _ptr(part), nparts, 2 * self.cout, 128, _ptr(folded), _stream())
                part, nparts = folded, 128
            e.stats_for[y.data_ptr()] = (part, nparts)
        return y

    def fwd_act(self, x, norm, temb=None, res=None, res2=None, want_raw=False):
        """conv(x) + bias (+ temb + res + res2) with the eval-mode BatchNorm `norm` + LeakyReLU of the consumer applied in
        the same launch: returns (activated, raw output or None), or None when no kernel covers the shape.  Bit-identical
        to fwd() followed by norm.fwd(..., training=False)."""
        B, H, W, _ = x.shape
        sp = self.plan(B, H, W)
        if self.up or not sp.act_ok:
            return None
        Ho, Wo = self.out_hw(H, W)
        coef = norm.eval_coef()
        y = torch.empty((B, Ho, Wo, self.cout), dtype=self.eng.dtype, device=x.device)
        raw = torch.empty_like(y) if want_raw else None
        ev = self._timed(0, B, H, W)
        call("vamd_conv2d_fwd_act", sp.ref, _ptr(x), _ptr(self.wf), _ptr(self.b), _ptr(temb), _ptr(res), _ptr(res2),
             _ptr(coef[0]), _ptr(coef[1]), norm.slope, _ptr(y), _ptr(raw), _stream())
        if ev is not None:
            ev.record()
        return y, raw

    def fwd_short(self, x, short, x2, res2=None, bias_sum=None, want_stats=False):
        """y = conv(x) + short(x2) + (both biases) (+ res2) in one launch (`short` is a 1x1 _Conv whose
        output has this conv's shape), or None when the kernels do not cover the shape."""
        B, H, W, _ = x.shape
        d = self.desc(B, H, W)
        if self.stride != 1 or self.up or not lib.vamd_conv2d_fwd_short_supported(C.byref(d), short.cin):
            return None
        e = self.eng
        y = torch.empty((B, H, W, self.cout), dtype=e.dtype, device=x.device)
        nparts = self.plan(B, H, W).parts if want_stats else 0
        part = torch.empty((nparts, self.cout, 2), dtype=torch.float32, device=x.device) if nparts > 0 else None
        ev = self._timed(0, B, H, W)
        call("vamd_conv2d_fwd_short", C.byref(d), _ptr(x), _ptr(self.wf), _ptr(bias_sum), None, _ptr(x2), _ptr(short.wf),
             short.cin, _ptr(res2), _ptr(y), _ptr(part), _stream())
        if ev is not None:
            ev.record()
        if nparts > 0:
            if nparts > 512:
                folded = torch.empty((128, self.cout, 2), dtype=torch.float32, device=x.device)
                call("vamd_fold_partials", _ptr(part), nparts, 2 * self.cout, 128, _ptr(folded), _stream())
                part, nparts = folded, 128
            e.stats_for[y.data_ptr()] = (part, nparts)
        return y

    def dgrad(self, dy, in_shape, add=None):
        B, H, W, _ = in_shape
        sp = self.plan(B, H, W)
        dx = torch.empty(in_shape, dtype=self.eng.dtype, device=dy.device)
        if sp.up4_dgrad:
            ev = self._timed(1, B, H, W)
            call("vamd_conv2d_dgrad_up4", sp.ref, _ptr(dy), _ptr(self.wd4), _ptr(add), _ptr(dx), _stream())
            if ev is not None:
                ev.record()
            return dx
        nbytes = sp.dgrad_ws
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device) if nbytes else None
        self.eng.side_keep(ws)
        ev = self._timed(1, B, H, W)
        call("vamd_conv2d_dgrad", sp.ref, _ptr(dy), _ptr(self.wd), _ptr(add), _ptr(dx), _ptr(ws),
             nbytes, _stream())
        if ev is not None:
            ev.record()
        return dx

    def wgrad(self, dy, x, bias=True, pre=None):
        """Weight gradient into the arena; with bias=True the bias gradient (column sums of dy)
        comes out of the same launch.  pre: as in fwd()."""
        with self.eng.side_section(dy, x, *(pre[:2] if pre is not None else ())):
            self._wgrad(dy, x, bias, pre)

    def _wgrad(self, dy, x, bias, pre):
        B, H, W, _ = x.shape
        gb = _ptr(self.gb) if bias else None
        ev = self._timed(2, B, H, W)
        sp = self.plan(B, H, W)
        nbytes = sp.wgrad_ws
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        self.eng.side_keep(ws)
        if pre is not None:
            call("vamd_conv2d_wgrad_prenorm_oihw", sp.ref, _ptr(dy), _ptr(x), _ptr(pre[0]), _ptr(pre[1]), pre[2], _ptr(self.gw),
                 gb, _ptr(ws), nbytes, _stream())
        else:
            call("vamd_conv2d_wgrad_oihw", sp.ref, _ptr(dy), _ptr(x), _ptr(self.gw), gb, _ptr(ws), nbytes, _stream())
        if ev is not None:
            ev.record()


# the q / k / v projections of an attention block as one 1x1 convolution over 3C output channels (forward, input
# gradient and weight gradient: three launches instead of nine, on a layer whose single projections fill a quarter of
# the chip)
FUSE_QKV = os.environ.get("VAMD_QKV_FUSE", "1") != "0"


class _QKV(_Conv):
    """q_proj, k_proj and v_proj (reference models/ddpm.py:95-97, applied to the same normalised input at :107-109) as
    ONE convolution: a filter [3C][C] whose row ranges are the three projections' packed weights (their `wf` become views
    of it), its transpose [C][3C] for the input gradient (written by the weight re-pack as three column ranges), the
    concatenated bias, and a weight gradient that lands in the three parameters' own .grad tensors."""

    def __init__(self, eng, q, k, v):
        Cc = q.cin
        self.eng, self.name, self.parts = eng, q.name + "|k|v", (q, k, v)
        self.cin, self.cout, self.k, self.stride, self.up, self.pad = Cc, 3 * Cc, 1, 1, False, 0
        self.wf = torch.empty((3 * Cc, 1, 1, Cc), dtype=eng.dtype, device=eng.device)
        self.wd = torch.empty((Cc, 1, 1, 3 * Cc), dtype=eng.dtype, device=eng.device)
        self.b = torch.empty(3 * Cc, dtype=torch.float32, device=eng.device)
        self.wd4 = self.wf4 = None
        self._descs, self._plans = {}, {}
        for j, c in enumerate(self.parts):
            c.wf = self.wf[j * Cc:(j + 1) * Cc]              # contiguous row range: the single projection still works
        self._dw3 = (C.c_void_p * 3)(*[c.gw.data_ptr() for c in self.parts])
        self._db3 = (C.c_void_p * 3)(*[c.gb.data_ptr() for c in self.parts])

    def repack_items(self):
        """(conv, fused input-gradient filter's column range, its row length, bias destination) per projection: what the
        engine's re-pack table needs besides the projections' own entries."""
        Cc = self.cin
        return [(c, self.wd.data_ptr() + j * Cc * self.wd.element_size(), 3 * Cc, self.b.data_ptr() + j * Cc * 4)
                for j, c in enumerate(self.parts)]

    def _wgrad(self, dy, x, bias, pre):
        B, H, W, _ = x.shape
        ev = self._timed(2, B, H, W)
        sp = self.plan(B, H, W)
        nbytes = sp.wgrad_ws
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        self.eng.side_keep(ws)
        call("vamd_conv2d_wgrad_oihw_split3", sp.ref, _ptr(dy), _ptr(x), C.byref(self._dw3), C.byref(self._db3) if bias else None,
             _ptr(ws), nbytes, _stream())
        if ev is not None:
            ev.record()


class _Linear:
    """A 1x1 conv applied to a [B, C, 1, 1] map (the time-embedding path), kept in fp32."""

    def __init__(self, eng, name, cin, cout, act_in):
        self.eng, self.cin, self.cout, self.act_in = eng, cin, cout, int(act_in)
        self.w, self.b = eng.param(name + ".weight"), eng.param(name + ".bias")
        self.gw, self.gb = eng.grad(name + ".weight"), eng.grad(name + ".bias")

    def fwd(self, x):
        y = torch.empty((x.shape[0], self.cout), dtype=torch.float32, device=x.device)
        call("vamd_linear_fwd", _ptr(x), _ptr(self.w), _ptr(self.b), _ptr(y), x.shape[0], self.cin,
             self.cout, self.act_in, LEAKY_SLOPE, _stream())
        return y

    def bwd(self, x, dy, dx=None, accumulate=False, dbias=True):
        call("vamd_linear_bwd", _ptr(x), _ptr(self.w), _ptr(dy), _ptr(dx), _ptr(self.gw),
             _ptr(self.gb) if dbias else None, x.shape[0], self.cin, self.cout, self.act_in, LEAKY_SLOPE,
             int(accumulate), _stream())


FUSE_SHORTCUT = os.environ.get("VAMD_FWD_SHORT", "0") != "0"
# eval-mode forward without a tape (the sampler): BatchNorm + LeakyReLU applied by the producing convolution's epilogue
FUSE_EVAL_NORM = os.environ.get("VAMD_EVAL_FUSE", "1") != "0"
# forwards without a tape: the attention core as one kernel (bf16, S <= 1024, C <= 512), bit-identical to the three
# separate kernels and without their S x S tensors in HBM.  Opt-in: measured 10-40 % slower than the three MFMA kernels
# at the sizes of the shipped configs (4 x 1024 x 256: 62 vs 49 us; 64 x 196 x 512: 95 vs 52 us; -0.7 % on ffhq sampling)
FUSE_ATTENTION = os.environ.get("VAMD_ATTN_FUSE", "0") != "0"
# backward: the gradients of every block's time-embedding projection in two launches at the end instead of two per block
BATCH_TEMB_BWD = os.environ.get("VAMD_TEMB_BWD_MULTI", "1") != "0"
# backward: weight gradients (with their reduce kernels) on a second stream.  Nothing on the input-gradient chain waits for
# them, so they run beside its memory-bound normalisation passes and fill the gaps between its dependent launches; they are
# joined before backward returns.  A gradient bucket is handed to the exchange behind both streams' work so far.
SIDE_WGRAD = os.environ.get("VAMD_SIDE_WGRAD", "1") != "0"


class _Norm:
    """Statistics + fused scale/shift/LeakyReLU(/dropout) of BatchNorm2d (G = 1) or GroupNorm (G = B)."""

    def __init__(self, eng, name, C_, batchnorm: bool, slope=None):
        self.eng, self.C, self.bn = eng, C_, batchnorm
        self.gamma, self.beta = eng.param(name + ".weight"), eng.param(name + ".bias")
        self.ggamma, self.gbeta = eng.grad(name + ".weight"), eng.grad(name + ".bias")
        if batchnorm:
            self.rm, self.rv = eng.buffer(name + ".running_mean"), eng.buffer(name + ".running_var")
            self.nbt = eng.buffer(name + ".num_batches_tracked")
        # the activation that follows the norm in the reference: LeakyReLU after a block's BatchNorm, none (slope 1)
        # after an attention block's norm
        self.slope = slope if slope is not None else (LEAKY_SLOPE if batchnorm else 1.0)
        self._eval_coef, self._eval_key = None, None

    def eval_coef(self):
        """(scale, shift, mean, rstd) [4][1][C] of an eval-mode BatchNorm: they depend on parameters and running
        statistics only, so they are kept until either changes (a sampler asks once per denoising step)."""
        e = self.eng
        key = (e.coef_epoch, e.flat_b._version)
        if self._eval_coef is None or self._eval_key != key:
            coef = torch.empty((4, 1, self.C), dtype=torch.float32, device=e.device)
            call("vamd_bn_finalize", None, 0, self.C, 0, _ptr(self.gamma), _ptr(self.beta), _ptr(self.rm),
                 _ptr(self.rv), None, BN_MOMENTUM, BN_EPS, 0, _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]), _ptr(coef[3]),
                 _stream())
            self._eval_coef, self._eval_key = coef, key
        return self._eval_coef

    def fwd(self, x, training, mask=None, mask_scale=1.0, apply=True):
        """(normalised + activated x, coefficients).  apply=False: the coefficients alone (the consumer normalises x
        itself); the first element is then None."""
        B, H, W, Cc = x.shape
        e, dev = self.eng, x.device
        G, R = (1, B * H * W) if self.bn else (B, H * W)
        if self.bn and not training:
            coef = self.eval_coef()
            sc, sh = coef[0], coef[1]
        else:
            coef = torch.empty((4, G, Cc), dtype=torch.float32, device=dev)   # scale, shift, mean, rstd
            sc, sh, mu, rs = coef[0], coef[1], coef[2], coef[3]
            fused = e.stats_for.pop(x.data_ptr(), None) if self.bn else None
            if fused is not None:                 # the producing convolution already summed y and y*y
                part, nchunk = fused
            else:
                nchunk = lib.vamd_stats_chunks(R, G)
                part = torch.empty((G, nchunk, Cc, 2), dtype=torch.float32, device=dev)
                call("vamd_chan_stats", _ptr(x), _ptr(part), G, R, Cc, e.dt, _stream())
            if self.bn:
                call("vamd_bn_finalize", _ptr(part), nchunk, Cc, R, _ptr(self.gamma), _ptr(self.beta),
                     _ptr(self.rm), _ptr(self.rv), _ptr(self.nbt), BN_MOMENTUM, BN_EPS, 1, _ptr(sc),
                     _ptr(sh), _ptr(mu), _ptr(rs), _stream())
            else:
                call("vamd_gn_finalize", _ptr(part), nchunk, B, Cc, GN_GROUPS, H * W, _ptr(self.gamma),
                     _ptr(self.beta), GN_EPS, _ptr(sc), _ptr(sh), _ptr(mu), _ptr(rs), _stream())
        if not apply:
            return None, coef
        y = torch.empty_like(x)
        call("vamd_scale_shift_act", _ptr(x), _ptr(sc), _ptr(sh), _ptr(mask), self.slope, mask_scale, _ptr(y), G, R,
             Cc, e.dt, _stream())
        return y, coef

    def bwd(self, dy, x, coef, mask=None, add=None, add2=None, need_dx=True, img_sums=None, training=True,
            mask_scale=1.0):
        """img_sums = (dsum [B][C], total [C] or None): BatchNorm only; filled with the per-image channel
        sums of dx (and their sum over the batch) from the statistics pass, without reading dx.
        training=False (BatchNorm only): the forward used the running statistics, which do not depend on
        x, so the mean / variance terms of the input gradient vanish."""
        fixed = int(self.bn and not training)
        B, H, W, Cc = x.shape
        e, dev = self.eng, x.device
        G, R = (1, B * H * W) if self.bn else (B, H * W)
        sc, sh, mu, rs = coef[0], coef[1], coef[2], coef[3]
        abc = torch.empty((3, G, Cc), dtype=torch.float32, device=dev)
        if img_sums is not None:
            nchunk = lib.vamd_stats_chunks(H * W, B)
            part = torch.empty((B, nchunk, Cc, 3), dtype=torch.float32, device=dev)
            call("vamd_bn_bwd_stats_img", _ptr(dy), _ptr(x), _ptr(mask), _ptr(sc), _ptr(sh), _ptr(mu), _ptr(rs),
                 self.slope, mask_scale, _ptr(part), B, H * W, Cc, e.dt, _stream())
            call("vamd_bn_bwd_finalize_img", _ptr(part), nchunk, B, Cc, H * W, _ptr(sc), _ptr(self.ggamma),
                 _ptr(self.gbeta), _ptr(abc[0]), _ptr(abc[1]), _ptr(abc[2]), _ptr(img_sums[0]), _ptr(img_sums[1]),
                 fixed, _stream())
        else:
            nchunk = lib.vamd_stats_chunks(R, G)
            part = torch.empty((G, nchunk, Cc, 2), dtype=torch.float32, device=dev)
            call("vamd_norm_bwd_stats", _ptr(dy), _ptr(x), _ptr(mask), _ptr(sc), _ptr(sh), _ptr(mu), _ptr(rs),
                 self.slope, mask_scale, _ptr(part), G, R, Cc, e.dt, _stream())
        if img_sums is not None:
            pass
        elif self.bn:
            call("vamd_bn_bwd_finalize", _ptr(part), nchunk, Cc, R, _ptr(sc), _ptr(self.ggamma),
                 _ptr(self.gbeta), _ptr(abc[0]), _ptr(abc[1]), _ptr(abc[2]), fixed, _stream())
        else:
            call("vamd_gn_bwd_finalize", _ptr(part), nchunk, B, Cc, GN_GROUPS, H * W, _ptr(self.gamma),
                 _ptr(rs), _ptr(self.ggamma), _ptr(self.gbeta), _ptr(abc[0]), _ptr(abc[1]), _ptr(abc[2]),
                 _stream())
        if not need_dx:
            return None
        dx = torch.empty_like(x)
        call("vamd_norm_bwd_apply", _ptr(dy), _ptr(x), _ptr(mask), _ptr(sc), _ptr(sh), _ptr(mu), _ptr(rs),
             self.slope, mask_scale, _ptr(abc[0]), _ptr(abc[1]), _ptr(abc[2]), _ptr(add), _ptr(add2), _ptr(dx), G, R,
             Cc, e.dt, _stream())
        return dx


class _ResBlock:
    """reference models/ddpm.py:43-87."""

    def __init__(self, eng, spec: LayerSpec, tf: int):
        p = spec.prefix
        self.eng, self.spec = eng, spec
        self.bn1 = _Norm(eng, p + ".block1.0.0", spec.cin, True)
        self.conv1 = _Conv(eng, p + ".block1.0.2", spec.cin, spec.cout, 3)
        self.temb = _Linear(eng, p + ".linear_temb" + eng.linear_suffix, tf, spec.cout, act_in=True) if tf else None   # no time embedding: tf = 0
        self.bn2 = _Norm(eng, p + ".block2.0.0", spec.cout, True)
        self.conv2 = _Conv(eng, p + ".block2.0.3", spec.cout, spec.cout, 3)
        # the 1x1 shortcut: an nn.Conv2d (DDPM) or a channel-wise nn.Linear wrapped in a module (the VQ-VAE-UNet blocks);
        # both store [Cout][Cin] row-major
        self.short = _Conv(eng, p + eng.shortcut_key, spec.cin, spec.cout, 1) if spec.cin != spec.cout else None
        if self.short is not None:
            self.short.own_scratch = True            # its forward and input gradient run beside conv1's / conv2's
        self.convs = [c for c in (self.conv1, self.conv2, self.short) if c is not None]
        self._gb_stale = False                   # conv1's bias gradient holds an eval-mode value
        # nn.Dropout's 1/(1-p) stays an fp32 kernel argument; the mask tensor holds exact 0/1 values
        self.mask_scale = 1.0 / (1.0 - spec.dropout) if spec.dropout > 0.0 else 1.0
        # conv2's bias + the shortcut's, for the launch that computes both convolutions (Engine.refresh)
        self.bsum = torch.zeros(spec.cout, dtype=torch.float32, device=eng.device) if self.short is not None else None

    def refresh_bias(self):
        if self.bsum is not None and FUSE_SHORTCUT:       # only the one-launch form of conv2 + shortcut reads it
            torch.add(self.conv2.b, self.short.b, out=self.bsum)

    def final_conv(self):
        return self.conv2

    def fwd(self, x, temb, training, res2, tape, drop_mask=None, want_stats=False, next_norm=None):
        fuse = FUSE_EVAL_NORM and not training and tape is None
        a1 = self.eng.act_for.pop(x.data_ptr(), None) if fuse else None   # left by the producer of x, if it could
        coef1 = None
        B, H, W, _ = x.shape
        # conv1 / conv2 normalise their input themselves where forward and weight gradient both can (a1 / a2 stay None)
        pre1 = not fuse and self.conv1.plan(B, H, W).prenorm
        pre2 = not fuse and not (training and self.spec.dropout > 0.0) and self.conv2.plan(B, H, W).prenorm
        if a1 is None:
            a1, coef1 = self.bn1.fwd(x, training, apply=not pre1)
        # projected with the other blocks' in one launch, if this is the embedding that launch projected
        tproj = self.eng.tproj_for.get(id(self)) if (self.temb is not None and self.eng.tproj_src is temb) else None
        if tproj is None and self.temb is not None:
            tproj = self.temb.fwd(temb)
        # the 1x1 shortcut reads x alone: issued now on its own stream, it runs (memory-bound) beside conv1 and the
        # normalisation in front of conv2, whose epilogue adds it; main waits for it right before that launch
        e = self.eng
        sc = x
        early = self.short is not None and not (FUSE_SHORTCUT and not fuse and not pre2)
        if early:
            with e.side_section(x, lane=1):
                sc = self.short.fwd(x)
        if fuse:
            # no tape: h1 itself is never needed, its normalised form comes straight out of conv1
            got = self.conv1.fwd_act(a1, self.bn2, temb=tproj)
            a2 = got[0] if got is not None else self.bn2.fwd(self.conv1.fwd(a1, temb=tproj), False)[0]
            e.side_wait(1)
            got = self.conv2.fwd_act(a2, next_norm, res=sc, res2=res2, want_raw=True) if next_norm is not None else None
            if got is None:
                return self.conv2.fwd(a2, res=sc, res2=res2)
            self.eng.act_for[got[1].data_ptr()] = got[0]
            return got[1]
        if pre1:
            h1 = self.conv1.fwd(x, temb=tproj, want_stats=training, pre=(coef1[0], coef1[1], self.bn1.slope))
        else:
            h1 = self.conv1.fwd(a1, temb=tproj, want_stats=training)
        mask = None
        if training and self.spec.dropout > 0.0:
            mask = drop_mask if drop_mask is not None else self.eng.draw_dropout(h1.shape, self.spec.dropout)
        a2, coef2 = self.bn2.fwd(h1, training, mask, self.mask_scale, apply=not pre2)
        y = None
        if pre2:
            e.side_wait(1)
            y = self.conv2.fwd(h1, res=sc, res2=res2, want_stats=want_stats, pre=(coef2[0], coef2[1], self.bn2.slope))
        # the 1x1 shortcut as extra reduction steps of conv2: measured +0.3 % on the training step and
        # -2.5 % on 16-image sampling batches against the two-launch form, so it is opt-in
        if y is None and self.short is not None and FUSE_SHORTCUT:
            y = self.conv2.fwd_short(a2, self.short, x, res2=res2, bias_sum=self.bsum, want_stats=want_stats)
        if y is None:
            if self.short is not None and not early:       # the one-launch form does not cover this shape after all
                sc = self.short.fwd(x)
            e.side_wait(1)
            y = self.conv2.fwd(a2, res=sc, res2=res2, want_stats=want_stats)
        if tape is not None:
            tape.append((self, (x, a1, coef1, h1, a2, coef2, mask, temb, training)))
        return y

    def bwd(self, saved, dy, extra, dtemb, need_dx=True):
        x, a1, coef1, h1, a2, coef2, mask, temb, training = saved
        e = self.eng
        # conv2 and shortcut share dy; their bias gradients are the same column sums of dy (copied: letting the shortcut's
        # weight-gradient launch form them again costs more than the copy launch, 19.29 vs 19.17 ms on the 224 x 224 step)
        # a2 is None where conv2 normalises h1 itself (forward above); likewise a1 / x for conv1
        # (one side section for the weight gradients that read dy: every section costs the host an event record and a wait)
        with e.side_section(dy, h1, a2, x, coef2, torch_ops=self.short is not None):
            if a2 is None:
                self.conv2._wgrad(dy, h1, True, (coef2[0], coef2[1], self.bn2.slope))
            else:
                self.conv2._wgrad(dy, a2, True, None)
            if self.short is not None:
                self.short.gb.copy_(self.conv2.gb)         # behind conv2's weight-gradient launch, wherever that runs
                self.short._wgrad(dy, x, False, None)
        dxs = dy
        if self.short is not None and need_dx:                 # beside conv2's input gradient; needed by bn1's backward
            with e.side_section(dy, lane=1):
                dxs = self.short.dgrad(dy, x.shape)
        elif self.short is not None:
            dxs = None
        da2 = self.conv2.dgrad(dy, h1.shape)
        # d(projected temb)[b][c] = sum over pixels of dh1; conv1 bias gradient = its sum over b: both
        # come out of the BatchNorm backward's statistics pass (dh1 is affine in the summed terms)
        if self.temb is not None:
            # with a slot in the step's shared buffer the projection's own gradients wait for ONE launch over all blocks
            # at the end of backward (Engine._temb_bwd_flush); else they are computed here
            dtproj = e.tbwd_for.get(id(self))
            batched = dtproj is not None
            if not batched:
                dtproj = torch.empty((dy.shape[0], self.spec.cout), dtype=torch.float32, device=dy.device)
            dh1 = self.bn2.bwd(da2, h1, coef2, mask, img_sums=(dtproj, self.conv1.gb), training=training,
                               mask_scale=self.mask_scale)
            if not batched:
                self.temb.bwd(temb, dtproj, dx=dtemb, accumulate=True)
        else:
            dh1 = self.bn2.bwd(da2, h1, coef2, mask, training=training, mask_scale=self.mask_scale)
        # conv1's bias feeds a train-mode BatchNorm, whose mean subtraction cancels it: the
        # gradient is exactly zero (dh1 sums to zero over the batch) and is not computed
        if training and self._gb_stale:
            self.conv1.gb.zero_()
        self._gb_stale = not training
        if a1 is None:
            self.conv1.wgrad(dh1, x, bias=not training, pre=(coef1[0], coef1[1], self.bn1.slope))
        else:
            self.conv1.wgrad(dh1, a1, bias=not training)
        da1 = self.conv1.dgrad(dh1, x.shape)
        e.side_wait(1)
        return self.bn1.bwd(da1, x, coef1, None, add=dxs, add2=extra, need_dx=need_dx, training=training)


class _AttnBlock:
    """reference models/ddpm.py:89-123: GroupNorm, q/k/v 1x1 convs, single-head softmax attention
    over the H*W positions, output 1x1 conv, residual."""

    def __init__(self, eng, spec: LayerSpec):
        p = spec.prefix
        self.eng, self.spec, self.C = eng, spec, spec.cout
        self.norm = _Norm(eng, p + ".norm", self.C, False)
        self.q = _Conv(eng, p + ".q_proj", self.C, self.C, 1)
        self.k = _Conv(eng, p + ".k_proj", self.C, self.C, 1)
        self.v = _Conv(eng, p + ".v_proj", self.C, self.C, 1)
        self.o = _Conv(eng, p + ".linear", self.C, self.C, 1)
        self.convs = [self.q, self.k, self.v, self.o]
        self.qkv = _QKV(eng, self.q, self.k, self.v) if FUSE_QKV else None

    # Rows of the [S x S] score / probability matrices are padded to Sp = S rounded up to 8
    # elements (16 bytes of bf16) with zeros, so every row is a whole number of 16-byte pieces.
    def _bmm_nt(self, A, Bm, out, batch, M, N, K, lda, ldb, ldc, alpha):
        e = self.eng
        call("vamd_bmm_nt", _ptr(A), _ptr(Bm), _ptr(out), batch, M, N, K, lda, ldb, ldc, alpha, e.dt,
             int(out.dtype == torch.float32), e.algo, _stream())
        return out

    def _bmm_tn(self, A, Bm, out, batch, M, N, K, lda, ldb, ldc):
        e = self.eng
        call("vamd_bmm_tn", _ptr(A), _ptr(Bm), _ptr(out), batch, M, N, K, lda, ldb, ldc, 1.0, e.dt, e.algo, _stream())
        return out

    def _transpose_padded(self, t, batch, R, Cc, ld, ldx=None):
        """t [batch][R][Cc] (rows ldx apart: t may be a channel range of a wider tensor) -> [batch][Cc][ld], zero padding."""
        out = torch.zeros((batch, Cc, ld), dtype=t.dtype, device=t.device)
        call("vamd_transpose_ld", _ptr(t), _ptr(out), batch, R, Cc, ldx or Cc, ld, self.eng.dt, _stream())
        return out

    def fwd(self, x, temb, training, res2, tape, drop_mask=None, want_stats=False, next_norm=None):
        B, H, W, Cc = x.shape
        S = H * W
        Sp = (S + 7) // 8 * 8
        e, dev = self.eng, x.device
        hn, coef = self.norm.fwd(x, training)
        fuse_core = tape is None and FUSE_ATTENTION and lib.vamd_attention_fwd_supported(S, Cc, e.dt)
        # (the fused form needs the MFMA weight-gradient kernel, which writes the three parameters' gradients separately)
        if self.qkv is not None and not fuse_core and self.qkv.plan(B, H, W).mfma[2]:
            qkv = self.qkv.fwd(hn)                            # [B][H][W][3C]: q, k, v are its channel ranges (rows 3C apart)
            q, k, v, ld = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], 3 * Cc
        else:
            q, k, v, ld = self.q.fwd(hn), self.k.fwd(hn), self.v.fwd(hn), Cc
        if fuse_core:
            # no backward to follow: scores, softmax and the product with the values in one kernel, nothing S x S in HBM
            vT = self._transpose_padded(v, B, S, Cc, Sp)
            o = torch.empty((B, H, W, Cc), dtype=e.dtype, device=dev)
            call("vamd_attention_fwd", _ptr(q), _ptr(k), _ptr(vT), _ptr(o), B, S, Cc, Sp, Cc ** -0.5, e.dt, _stream())
            got = self.o.fwd_act(o, next_norm, res=x, res2=res2, want_raw=True) if next_norm is not None else None
            if got is not None:
                e.act_for[got[1].data_ptr()] = got[0]
                return got[1]
            return self.o.fwd(o, res=x, res2=res2)
        scores = torch.empty((B, S, Sp), dtype=torch.float32, device=dev)     # [B][query][key]
        self._bmm_nt(q, k, scores, B, S, S, Cc, ld, ld, Sp, Cc ** -0.5)
        P = torch.zeros((B, S, Sp), dtype=e.dtype, device=dev)
        call("vamd_softmax_rows", _ptr(scores), _ptr(P), B * S, S, Sp, e.dt, _stream())
        vT = self._transpose_padded(v, B, S, Cc, Sp, ldx=ld)                  # [B][C][Sp]
        o = torch.empty((B, H, W, Cc), dtype=e.dtype, device=dev)             # o[s][c] = sum_t P[s][t] v[t][c]
        self._bmm_nt(P, vT, o, B, S, Cc, S, Sp, Sp, Cc, 1.0)
        got = self.o.fwd_act(o, next_norm, res=x, res2=res2, want_raw=True) if next_norm is not None else None
        if got is not None:                       # the next block's eval-mode BatchNorm rides on the output projection
            e.act_for[got[1].data_ptr()] = got[0]
            return got[1]
        y = self.o.fwd(o, res=x, res2=res2, want_stats=want_stats)
        if tape is not None:
            tape.append((self, (x, hn, coef, q, k, v, P, o)))
        return y

    def bwd(self, saved, dy, extra, dtemb, need_dx=True):
        x, hn, coef, q, k, v, P, o = saved
        e, dev = self.eng, x.device
        B, H, W, Cc = x.shape
        S = H * W
        Sp = P.shape[2]
        self.o.wgrad(dy, o)
        do = self.o.dgrad(dy, o.shape)
        ld = q.stride(2)                                                      # 3C when q, k, v are ranges of one tensor
        fused = self.qkv is not None and ld == 3 * Cc
        if fused:
            dqkv = torch.empty((B, H, W, 3 * Cc), dtype=e.dtype, device=dev)
            dq, dk, dv, ldg = dqkv[..., :Cc], dqkv[..., Cc:2 * Cc], dqkv[..., 2 * Cc:], 3 * Cc
        else:
            dq, dk, dv = (torch.empty((B, H, W, Cc), dtype=e.dtype, device=dev) for _ in range(3))
            ldg = Cc
        dP = torch.empty((B, S, Sp), dtype=torch.float32, device=dev)         # dP[s][t] = sum_c do[s][c] v[t][c]
        self._bmm_nt(do, v, dP, B, S, S, Cc, Cc, ld, Sp, 1.0)
        self._bmm_tn(P, do, dv, B, S, Cc, S, Sp, Cc, ldg)                     # dv[t][c] = sum_s P[s][t] do[s][c]
        dS = torch.zeros((B, S, Sp), dtype=e.dtype, device=dev)
        call("vamd_softmax_rows_bwd", _ptr(P), _ptr(dP), _ptr(dS), B * S, S, Sp, Cc ** -0.5, e.dt, _stream())
        kT = self._transpose_padded(k, B, S, Cc, Sp, ldx=ld)
        self._bmm_nt(dS, kT, dq, B, S, Cc, S, Sp, Sp, ldg, 1.0)               # dq[s][c] = sum_t dS[s][t] k[t][c]
        self._bmm_tn(dS, q, dk, B, S, Cc, S, Sp, ld, ldg)                     # dk[t][c] = sum_s dS[s][t] q[s][c]
        if fused:
            self.qkv.wgrad(dqkv, hn)
            dhn = self.qkv.dgrad(dqkv, hn.shape)
        else:
            dhn = None
            for conv, g in ((self.q, dq), (self.k, dk), (self.v, dv)):
                conv.wgrad(g, hn)
                dhn = conv.dgrad(g, hn.shape, add=dhn)
        return self.norm.bwd(dhn, x, coef, None, add=dy, add2=extra, need_dx=need_dx)


class _ChanAttnBlock(_AttnBlock):
    """The attention block of the reference's latent-diffusion and first-stage files (models/ldm.py:91-108,
    models/vqvaeunet.py:83-100): BatchNorm2d without an activation, ONE channel-wise Linear whose output serves as
    query, key and value, attention BETWEEN CHANNELS (scores [B, C, C] = p p^T * C^-0.5 over the H*W positions, softmax
    over the second channel axis), a second Linear, residual.  With p stored [B][S][C]:
        scores = pT pT^T (pT = p transposed, rows padded to 8), W = softmax(scores), o = p W^T, y = x + linear2(o)."""

    def __init__(self, eng, spec: LayerSpec):
        p = spec.prefix
        self.eng, self.spec, self.C = eng, spec, spec.cout
        self.norm = _Norm(eng, p + ".norm", self.C, True, slope=1.0)
        self.lin1 = _Conv(eng, p + ".linear1.linear", self.C, self.C, 1)
        self.lin2 = _Conv(eng, p + ".linear2.linear", self.C, self.C, 1)
        self.convs = [self.lin1, self.lin2]

    def fwd(self, x, temb, training, res2, tape, drop_mask=None, want_stats=False, next_norm=None):
        B, H, W, Cc = x.shape
        S = H * W
        Sp = (S + 7) // 8 * 8
        e, dev = self.eng, x.device
        hn, coef = self.norm.fwd(x, training)
        p = self.lin1.fwd(hn)
        pT = self._transpose_padded(p, B, S, Cc, Sp)                          # [B][C][Sp]
        scores = torch.empty((B, Cc, Cc), dtype=torch.float32, device=dev)
        self._bmm_nt(pT, pT, scores, B, Cc, Cc, S, Sp, Sp, Cc, Cc ** -0.5)
        Wt = torch.empty((B, Cc, Cc), dtype=e.dtype, device=dev)
        call("vamd_softmax_rows", _ptr(scores), _ptr(Wt), B * Cc, Cc, Cc, e.dt, _stream())
        o = torch.empty((B, H, W, Cc), dtype=e.dtype, device=dev)             # o[s][c1] = sum_c2 W[c1][c2] p[s][c2]
        self._bmm_nt(p, Wt, o, B, S, Cc, Cc, Cc, Cc, Cc, 1.0)
        got = self.lin2.fwd_act(o, next_norm, res=x, res2=res2, want_raw=True) if next_norm is not None else None
        if got is not None:
            e.act_for[got[1].data_ptr()] = got[0]
            return got[1]
        y = self.lin2.fwd(o, res=x, res2=res2, want_stats=want_stats)
        if tape is not None:
            tape.append((self, (x, hn, coef, p, pT, Wt, o, training)))
        return y

    def bwd(self, saved, dy, extra, dtemb, need_dx=True):
        x, hn, coef, p, pT, Wt, o, training = saved
        e, dev = self.eng, x.device
        B, H, W, Cc = x.shape
        S = H * W
        Sp = pT.shape[2]
        self.lin2.wgrad(dy, o)
        do = self.lin2.dgrad(dy, o.shape)
        doT = self._transpose_padded(do, B, S, Cc, Sp)
        dW = torch.empty((B, Cc, Cc), dtype=torch.float32, device=dev)        # dW[c1][c2] = sum_s do[s][c1] p[s][c2]
        self._bmm_nt(doT, pT, dW, B, Cc, Cc, S, Sp, Sp, Cc, 1.0)
        dS = torch.empty((B, Cc, Cc), dtype=e.dtype, device=dev)
        call("vamd_softmax_rows_bwd", _ptr(Wt), _ptr(dW), _ptr(dS), B * Cc, Cc, Cc, Cc ** -0.5, e.dt, _stream())
        # dp[s][c] = sum_c1 do[s][c1] W[c1][c] + sum_c2 dS[c][c2] p[s][c2] + sum_c1 dS[c1][c] p[s][c1]: one product
        # [do | p | p] x [W^T | dS | dS^T]^T over 3C, so the three terms meet in the fp32 accumulators
        lhs = torch.cat([do.view(B, S, Cc), p.view(B, S, Cc), p.view(B, S, Cc)], dim=2)
        rhs = torch.cat([self._transpose_padded(Wt, B, Cc, Cc, Cc), dS, self._transpose_padded(dS, B, Cc, Cc, Cc)], dim=2)
        dp = torch.empty((B, H, W, Cc), dtype=e.dtype, device=dev)
        self._bmm_nt(lhs, rhs, dp, B, S, Cc, 3 * Cc, 3 * Cc, 3 * Cc, Cc, 1.0)
        self.lin1.wgrad(dp, hn)
        dhn = self.lin1.dgrad(dp, hn.shape)
        return self.norm.bwd(dhn, x, coef, None, add=dy, add2=extra, need_dx=need_dx, training=training)


class _Resample:
    """DownBlock (conv 3x3 stride 2, or AvgPool2d((2, 2)) when the model is built with with_conv false) /
    UpBlock (nearest 2x upsample, fused into the conv 3x3 that follows it when there is one);
    reference models/ddpm.py:14-41."""

    def __init__(self, eng, spec: LayerSpec):
        p = spec.prefix
        self.eng, self.spec = eng, spec
        self.down = spec.kind == "down"
        if not eng.plan.with_conv:
            self.conv = None
        elif self.down:
            self.conv = _Conv(eng, p + ".layers.0", spec.cin, spec.cout, 3, stride=2)
        else:
            self.conv = _Conv(eng, p + ".layers.1", spec.cin, spec.cout, 3, up=True)
        self.convs = [self.conv] if self.conv is not None else []

    def _pool(self, x, add, scale):
        B, H, W, Cc = x.shape
        y = torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        call("vamd_pool2x2", _ptr(x), _ptr(add), _ptr(y), B, H, W, Cc, scale, self.eng.dt, _stream())
        return y

    def _unpool(self, x, add, scale, H, W):
        B, Hs, Ws, Cc = x.shape
        y = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
        call("vamd_unpool2x2", _ptr(x), _ptr(add), _ptr(y), B, Hs, Ws, Cc, H, W, scale, self.eng.dt, _stream())
        return y

    def fwd(self, x, temb, training, res2, tape, drop_mask=None, want_stats=False, next_norm=None):
        if self.conv is not None:
            got = self.conv.fwd_act(x, next_norm, res2=res2, want_raw=True) if next_norm is not None else None
            if got is not None:
                self.eng.act_for[got[1].data_ptr()] = got[0]
                return got[1]
            y = self.conv.fwd(x, res2=res2, want_stats=want_stats)
        elif self.down:
            y = self._pool(x, res2, 0.25)
        else:
            y = self._unpool(x, res2, 1.0, 2 * x.shape[1], 2 * x.shape[2])
        if tape is not None:
            tape.append((self, (x,) if self.conv is not None else (x.shape,)))
        return y

    def bwd(self, saved, dy, extra, dtemb, need_dx=True):
        if self.conv is not None:
            (x,) = saved
            self.conv.wgrad(dy, x)
            return self.conv.dgrad(dy, x.shape, add=extra)
        (shape,) = saved
        if self.down:
            return self._unpool(dy, extra, 0.25, shape[1], shape[2])
        return self._pool(dy, extra, 1.0)


class Engine:
    # parameter-key conventions of the reference file the model comes from (models/ddpm.py: 1x1 nn.Conv2d everywhere)
    shortcut_key = ".layers.0"           # a block's 1x1 shortcut
    linear_suffix = ""                   # time-embedding linears, their per-block projections and the End projection
    attn_block = _AttnBlock

    def __init__(self, module: torch.nn.Module, plan: UNetPlan, dtype: torch.dtype, algo=_lib.ALGO_AUTO):
        self._init_arenas(module, plan, dtype, algo)
        self._init_layers(plan)

    def _init_arenas(self, module, plan, dtype, algo):
        """Parameters, gradients and BatchNorm buffers of `module` become views of flat arenas; launch-independent state."""
        params = list(module.named_parameters())
        if not params or params[0][1].device.type != "cuda":
            raise RuntimeError("vision_amd needs its parameters on a HIP device (cuda:N); there is no CPU path")
        self.device = params[0][1].device
        self.plan, self.dtype, self.algo = plan, dtype, algo
        self.stats_for: Dict[int, tuple] = {}      # output data_ptr -> (partial sums, parts) left by a conv epilogue
        self.act_for: Dict[int, torch.Tensor] = {}  # output data_ptr -> its eval-mode normalised form (same launch)
        # zero-filled once, then owned by the kernels (flags and partial sums of tiles split between workgroups)
        # (partial products of layers whose reduction is split between workgroups: the small, deep layers)
        self.scratch = torch.zeros(_lib.CONV_SCRATCH_BYTES, dtype=torch.uint8, device=self.device)
        self.scratch_aux = torch.zeros(_lib.CONV_SCRATCH_BYTES, dtype=torch.uint8, device=self.device)   # the shortcuts'
        self.dt = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}[dtype]

        # one flat fp32 arena for parameters and one for gradients; every nn.Parameter becomes a view
        sizes = [p.numel() for _, p in params]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 3) // 4 * 4              # keep every tensor 16-byte aligned
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._p, self._g = {}, {}
        self.param_slices = {}
        with torch.no_grad():
            for (name, p), off, n in zip(params, offs, sizes):
                view = self.flat_p[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                g = self.flat_g[off:off + n].view(p.shape)
                p.grad = g
                self._p[name], self._g[name] = view, g
                self.param_slices[name] = (off, n)
        # BatchNorm buffers likewise: one fp32 arena (running_mean / running_var) and one int64
        # arena (num_batches_tracked), so a data-parallel buffer broadcast is two collectives
        bufs = list(module.named_buffers())
        fl = [(n, b) for n, b in bufs if b.dtype == torch.float32]
        it = [(n, b) for n, b in bufs if b.dtype == torch.int64]
        assert len(fl) + len(it) == len(bufs), "unexpected buffer dtype"
        self.flat_b = torch.zeros(sum((b.numel() + 3) // 4 * 4 for _, b in fl), dtype=torch.float32, device=self.device)
        self.flat_nbt = torch.zeros(max(1, len(it)), dtype=torch.int64, device=self.device)
        self._b = {}
        with torch.no_grad():
            off = 0
            for n, b in fl:
                view = self.flat_b[off:off + b.numel()].view(b.shape)
                view.copy_(b)
                self._rebind_buffer(module, n, view)
                off += (b.numel() + 3) // 4 * 4
            for i, (n, b) in enumerate(it):
                view = self.flat_nbt[i:i + 1].view(b.shape)
                view.copy_(b)
                self._rebind_buffer(module, n, view)
        self._params = [p for _, p in params]
        self.reducer = None               # set by vision_amd.parallel.DataParallel
        self._versions = None
        self.coef_epoch = 0
        self.dirty = True
        self._repack_table = None
        self.drop_generator = None
        self.profile = None               # list -> receives (kind, flops, start, end) per MFMA conv launch
        self.side_wgrad = SIDE_WGRAD        # weight gradients of backward on a second stream (see SIDE_WGRAD)
        self._side_streams, self._side_raw, self._side_used, self._side_on, self._side_keep = None, None, [False, False], False, []
        self.debug_taps = None            # dict -> receives (output, folded skip) per layer (diagnostics)
        self._tproj_cache, self.tproj_for, self.tproj_src = {}, {}, None
        self._tbwd_cache, self.tbwd_for = {}, {}

    def _init_layers(self, plan):
        tf = plan.time_features
        self.lin1 = _Linear(self, "linear_1" + self.linear_suffix, plan.ch, tf, act_in=False)
        self.lin2 = _Linear(self, "linear_2" + self.linear_suffix, tf, tf, act_in=True)
        self.layers = []
        for spec in plan.layers:
            if spec.kind == "res":
                self.layers.append(_ResBlock(self, spec, tf))
            elif spec.kind == "attn":
                self.layers.append(self.attn_block(self, spec))
            else:
                self.layers.append(_Resample(self, spec))
        self.end = _Conv(self, "End.0.0" + self.linear_suffix, plan.end_channels, plan.end_channels, 1)
        self.n_down = len(plan.section("Downsampling"))
        self.n_mid = len(plan.section("Middle"))
        # 2*pi*exp(linspace(ln 1, ln 1000, ch/2)) tabulated with the reference's own op sequence
        # (models/ddpm.py:126-134) on the host, once
        freqs = torch.exp(torch.linspace(math.log(1.0), math.log(1000.0), plan.ch // 2))
        self.angular = (2.0 * math.pi * freqs).to(self.device)
        self._lows = {spec.prefix: self._layer_low(spec.prefix) for spec in plan.layers}
        self._lows["End"] = self._layer_low("End")

    def _rebind_buffer(self, module, name, view):
        owner, _, leaf = name.rpartition(".")
        sub = module.get_submodule(owner) if owner else module
        sub._buffers[leaf] = view
        self._b[name] = view

    def layer_offsets(self):
        """Arena offsets at which each layer's parameters start (creation order = arena order)."""
        offs = set()
        for name, (off, n) in self.param_slices.items():
            parts = name.split(".")
            key = ".".join(parts[:2]) if parts[0] in ("Downsampling", "Middle", "Upsampling") else parts[0]
            offs.add((key, off))
        first = {}
        for key, off in offs:
            first[key] = min(off, first.get(key, off))
        return sorted(first.values())

    def _layer_low(self, prefix):
        """Lowest arena offset of the layer's parameters; None for a layer without any (pooling / bare upsampling)."""
        return min((off for name, (off, n) in self.param_slices.items() if name.startswith(prefix + ".")), default=None)

    # ---- parameter access for the layer objects
    def param(self, name):
        return self._p[name]

    def grad(self, name):
        return self._g[name]

    def buffer(self, name):
        return self._b[name]

    def owns(self, module) -> bool:
        p = next(module.parameters())
        if p.data_ptr() != self.flat_p.data_ptr():
            return False
        if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr():
            self.attach_grads()
        return True

    def all_convs(self):
        for layer in self.layers:
            yield from layer.convs
        yield self.end

    def refresh(self):
        """Re-pack the compute-dtype weight copies if any master weight changed."""
        vers = [p._version for p in self._params]
        if self.dirty or vers != self._versions:
            if self._repack_table is None:                    # device table of every conv, built once
                # (conv, input-gradient filter, its row length or 0, bias destination or 0); the fused q / k / v projections
                # of the attention blocks add one entry per projection for their side-by-side input-gradient filter
                entries = [(c, c.wd.data_ptr(), 0, 0) for c in self.all_convs()]
                for layer in self.layers:
                    if getattr(layer, "qkv", None) is not None:
                        entries += layer.qkv.repack_items()
                items = (RepackItem * len(entries))()
                blocks = 0
                for it, (c, wd, ld, bias_out) in zip(items, entries):
                    it.w, it.w_fwd, it.w_dgrad, it.ld_dgrad = c.w.data_ptr(), c.wf.data_ptr(), wd, ld
                    it.first_block, it.Cout, it.Cin, it.ksize = blocks, c.cout, c.cin, c.k
                    if bias_out:
                        it.bias, it.bias_out = c.b.data_ptr(), bias_out
                    if wd == c.wd.data_ptr() and c.wd4 is not None:       # an upsampling convolution's combined filters
                        it.w_up4_dgrad, it.w_up4_fwd = c.wd4.data_ptr(), c.wf4.data_ptr()
                    blocks += ((c.cout + 31) // 32) * ((c.cin + 31) // 32)
                raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
                self._repack_table = (raw, len(entries), blocks)
            raw, n, blocks = self._repack_table
            call("vamd_repack_weights", _ptr(raw), n, blocks, self.dt, _stream())
            for layer in self.layers:
                if isinstance(layer, _ResBlock):
                    layer.refresh_bias()
            self._versions = vers
            self.dirty = False
            self.coef_epoch += 1                              # parameters changed: cached eval-mode BatchNorm coefficients are stale

    def _project_temb(self, temb, B):
        """Every ResNetBlock's projection of the step's time embedding (reference models/ddpm.py:77) in one
        launch, into a buffer kept per batch size (the blocks read their slice as conv1's epilogue term;
        nothing of it is needed by backward)."""
        blocks = [l for l in self.layers if isinstance(l, _ResBlock)]
        self.tproj_for, self.tproj_src = {}, None
        if len(blocks) < 2 or len(blocks) > 64:
            return
        ent = self._tproj_cache.get(B)
        if ent is None:
            total = sum(l.spec.cout for l in blocks)
            buf = torch.empty(B * total, dtype=torch.float32, device=self.device)
            items = (LinearItem * len(blocks))()
            views, first = [], 0
            for it, l in zip(items, blocks):
                v = buf[B * first:B * (first + l.spec.cout)].view(B, l.spec.cout)
                it.W, it.bias, it.y, it.O, it.first = l.temb.w.data_ptr(), l.temb.b.data_ptr(), v.data_ptr(), l.spec.cout, first
                views.append(v)
                first += l.spec.cout
            table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
            ent = (buf, table, views, total, blocks[0].temb.cin)
            self._tproj_cache = {B: ent}                      # one batch size at a time is kept
        buf, table, views, total, cin = ent
        call("vamd_linear_fwd_multi", _ptr(temb), _ptr(table), len(blocks), total, B, cin, 1, LEAKY_SLOPE, _stream())
        self.tproj_for = {id(l): v for l, v in zip(blocks, views)}
        self.tproj_src = temb

    def _temb_bwd_setup(self, B):
        """Slots [B][cout] for every ResNetBlock's gradient of its projected time embedding, in one buffer kept per
        batch size, and the device table of vamd_linear_bwd_multi over them."""
        blocks = [l for l in self.layers if isinstance(l, _ResBlock) and l.temb is not None]
        self.tbwd_for = {}
        # with a live gradient exchange a layer's gradients must be final when its backward returns (its bucket may be
        # reduced right away), so the projections' gradients stay with their blocks
        exchanging = self.reducer is not None and getattr(self.reducer, "_on", True)
        if not BATCH_TEMB_BWD or exchanging or len(blocks) < 2 or len(blocks) > 64:
            return None
        ent = self._tbwd_cache.get(B)
        if ent is None:
            total = sum(l.spec.cout for l in blocks)
            buf = torch.empty(B * total, dtype=torch.float32, device=self.device)
            items = (LinearBwdItem * len(blocks))()
            views, first, tiles = [], 0, 0
            for it, l in zip(items, blocks):
                v = buf[B * first:B * (first + l.spec.cout)].view(B, l.spec.cout)
                it.W, it.dy, it.dW, it.dbias, it.O = (l.temb.w.data_ptr(), v.data_ptr(), l.temb.gw.data_ptr(),
                                                      l.temb.gb.data_ptr(), l.spec.cout)
                views.append(v)
                first += l.spec.cout
                tiles += (l.spec.cout + 15) // 16
            table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
            ent = (buf, table, views, tiles, len(blocks), blocks[0].temb.cin)
            self._tbwd_cache = {B: ent}
        self.tbwd_for = {id(l): v for l, v in zip(blocks, ent[2])}
        return ent

    def _temb_bwd_flush(self, ent, temb, dtemb):
        buf, table, views, tiles, n, cin = ent
        call("vamd_linear_bwd_multi", _ptr(temb), _ptr(table), n, tiles, temb.shape[0], cin, 1, LEAKY_SLOPE, _ptr(dtemb), 0,
             _stream())
        self.tbwd_for = {}

    def draw_dropout(self, shape, p):
        """0/1 keep mask in the compute dtype, one launch; the kernels apply 1/(1-p) in fp32."""
        return torch.empty(shape, dtype=self.dtype, device=self.device).bernoulli_(1.0 - p, generator=self.drop_generator)

    # ---- UNet forward / backward on NHWC tensors
    def forward(self, x, t, training: bool, tape: Optional[list], drop_masks=None):
        """x: [B,H,W,3] compute dtype; t: [B] fp32 (the reference feeds noise_rate**2).
        Returns the NHWC prediction; if `tape` is a list it receives what backward needs."""
        self.refresh()
        self.stats_for.clear()
        self.act_for.clear()
        if training:
            self.coef_epoch += 1                              # running statistics are about to change
        self._side_begin(x.device)                            # the blocks' 1x1 shortcuts run beside their first convolution
        B = x.shape[0]
        drop_masks = drop_masks or {}
        emb = torch.empty((B, self.plan.ch), dtype=torch.float32, device=x.device)
        call("vamd_timeembed", _ptr(t), _ptr(self.angular), _ptr(emb), B, self.plan.ch, _stream())
        t1 = self.lin1.fwd(emb)
        temb = self.lin2.fwd(t1)
        if tape is not None:
            tape.append(("temb", (emb, t1, temb)))
        self._project_temb(temb, B)
        skips: List[torch.Tensor] = []
        h = x
        nd, nm = self.n_down, self.n_mid
        n_up = len(self.layers) - nd - nm
        for i, layer in enumerate(self.layers):
            # the reference adds skip[j] (reversed order) in place to the input of up module j
            # (models/ddpm.py:228-232); here the producer of that input adds it in its epilogue
            res2 = None
            j = i + 1 - nd - nm                     # index of the up module consuming this output
            if 0 <= j < n_up and j < nd:
                res2 = skips[nd - 1 - j]
            # a train-mode ResBlock next: its first BatchNorm takes its statistics from this layer's epilogue
            nxt = self.layers[i + 1] if i + 1 < len(self.layers) else None
            want = training and isinstance(nxt, _ResBlock)
            # an eval-mode ResBlock next and no backward to follow: its first BatchNorm + LeakyReLU is applied by this
            # layer's last convolution, which then writes both the plain output and the normalised one
            nn_ = nxt.bn1 if (FUSE_EVAL_NORM and not training and tape is None and isinstance(nxt, _ResBlock)) else None
            h = layer.fwd(h, temb, training, res2, tape, drop_masks.get(layer.spec.prefix), want_stats=want, next_norm=nn_)
            if self.debug_taps is not None:
                self.debug_taps[layer.spec.prefix] = (h, res2)
            if i < nd:
                skips.append(h)
        if nm == 0 and n_up > 0 and nd > 0:
            raise RuntimeError("unexpected plan without Middle section")
        out = self.end.fwd(h)
        if tape is not None:
            tape.append(("end", (h,)))
        self._side_join()
        return out

    @contextlib.contextmanager
    def side_section(self, *tensors, torch_ops=False, lane=0):
        """With the extra streams on (inside forward and backward): the library launches of the block go to stream `lane`
        (0: backward's weight gradients, joined when backward returns; 1: the 1x1 shortcut convolutions, which main waits
        for at side_wait(1)), behind everything the main stream has been given so far.  `tensors` (what the block reads)
        and whatever the block passes to side_keep() (workspaces it allocates: they come from the main stream's pool) stay
        referenced until the streams are joined, so the allocator cannot hand their memory to a later main-stream tensor.
        torch_ops: the block calls torch operators or collectives, so torch's current stream is switched too (a few
        microseconds of host time more).  Otherwise a no-op."""
        if not self._side_on:
            yield
            return
        call("vamd_stream_wait", self._side_raw[lane], _raw_stream(_cur_device()))
        self._side_used[lane] = True
        self._side_keep.extend(tensors)
        if torch_ops:
            with torch.cuda.stream(self._side_streams[lane]):
                yield
            return
        _override.raw = self._side_raw[lane]
        try:
            yield
        finally:
            _override.raw = None

    def side_keep(self, t):
        if self._side_on and t is not None:
            self._side_keep.append(t)

    def side_wait(self, lane):
        """The main stream waits for what stream `lane` has been given so far (a result computed there is needed now)."""
        if self._side_on and self._side_used[lane]:
            call("vamd_stream_wait", _raw_stream(_cur_device()), self._side_raw[lane])
            self._side_used[lane] = False

    def _side_begin(self, device):
        self._side_on = bool(self.side_wgrad) and device.type == "cuda"
        if self._side_on and self._side_streams is None:
            self._side_streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
            self._side_raw = [st.cuda_stream for st in self._side_streams]

    def _side_join(self):
        if self._side_on:
            for lane in (0, 1):
                self.side_wait(lane)
            self._side_on = False
        self._side_keep.clear()

    def backward(self, tape: list, dout):
        """dout: gradient of the NHWC prediction.  Writes every parameter gradient into the flat
        gradient arena (overwriting), returns nothing (the network input needs no gradient)."""
        tag, (h_end,) = tape.pop()
        assert tag == "end"
        red = self.reducer
        if red is not None:
            red.begin()
        self._side_begin(dout.device)
        self.end.wgrad(dout, h_end)
        g = self.end.dgrad(dout, h_end.shape)
        live = red is not None and getattr(red, "_on", True)      # ranks to exchange with: a bucket goes out behind both
        if red is not None:                                        # streams' work so far (else the call only keeps count)
            with self.side_section(torch_ops=True) if live else contextlib.nullcontext():
                red.layer_done(self._lows["End"])
        nd, nm = self.n_down, self.n_mid
        n_up = len(self.layers) - nd - nm
        B = dout.shape[0]
        dtemb = torch.zeros((B, self.plan.time_features), dtype=torch.float32, device=dout.device)
        tbwd = self._temb_bwd_setup(B)
        dskip: Dict[int, torch.Tensor] = {}
        for i in range(len(self.layers) - 1, -1, -1):
            layer, saved = tape.pop()
            assert layer is self.layers[i]
            j = i - nd - nm                         # up-module index (>= 0 inside Upsampling)
            # gradient reaching the output of layer i-1 = dx of layer i (+ the skip gradient if
            # layer i-1 is a down module whose output was also used as a skip)
            extra = dskip.pop(i - 1, None) if i - 1 < nd else None
            g_in = layer.bwd(saved, g, extra, dtemb, need_dx=(i > 0))
            if 0 <= j < nd:
                # input of up module j was h + skip[nd-1-j]: the same gradient also reaches that skip.
                # `extra` was already folded into g_in for layer i-1, so hand the skip the plain part:
                # for up modules i-1 >= nd, hence extra is None and g_in is exactly that gradient.
                dskip[nd - 1 - j] = g_in
            g = g_in
            if red is not None and self._lows[layer.spec.prefix] is not None:
                with self.side_section(torch_ops=True) if live else contextlib.nullcontext():
                    red.layer_done(self._lows[layer.spec.prefix])
        tag, (emb, t1, temb) = tape.pop()
        assert tag == "temb"
        if tbwd is not None:
            self._temb_bwd_flush(tbwd, temb, dtemb)
        dt1 = torch.empty_like(t1)
        self.lin2.bwd(t1, dtemb, dx=dt1)
        self.lin1.bwd(emb, dt1, dx=None)
        if red is not None:
            with self.side_section(torch_ops=True) if live else contextlib.nullcontext():
                red.finish()
        self._side_join()
        self.attach_grads()

    def attach_grads(self):
        """Make sure every parameter's .grad is its view of the gradient arena (an optimiser's
        zero_grad(set_to_none=True) detaches them)."""
        for (name, g), p in zip(self._g.items(), self._params):
            if p.grad is not g:
                p.grad = g


class LatentEngine(Engine):
    """The latent-space UNet of reference models/ldm.py: the DDPM graph with channel-wise nn.Linear modules (wrapped, key
    `.linear`) where models/ddpm.py has 1x1 convolutions, and the channel-attention block."""
    shortcut_key = ".layers.0.linear"
    linear_suffix = ".linear"
    attn_block = _ChanAttnBlock
