"""ctypes binding of libvision_amd.so (the C ABI declared in include/vision_amd.h).

The library is built in-tree by `python vision_amd/build_ext.py` (hipcc, gfx950).  There is no
fallback: if the shared object is missing, importing this module raises.
"""
import ctypes as C
import os

# torch must be imported first: its wheel bundles the HIP runtime (libamdhip64) that owns the
# device memory and streams handed to this library; loading libvision_amd.so first would pull a
# second copy of the runtime from /opt/rocm into the process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# VAMD_LIB: another build of the same library (side-by-side kernel experiments)
LIB_PATH = os.environ.get("VAMD_LIB") or os.path.join(_HERE, "lib", "libvision_amd.so")

F32, BF16 = 0, 1
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_THIN = 0, 1, 2, 3


class VamdError(RuntimeError):
    pass


_DEFAULT_SCRATCH = None      # (tensor, nbytes) attached to every new ConvDesc (set_default_scratch)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "H", "W", "Cin", "Cout", "ksize", "stride", "pad", "upsample", "dtype", "algo")] + \
               [("scratch", C.c_void_p), ("scratch_bytes", C.c_uint64)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if _DEFAULT_SCRATCH is not None and not self.scratch:
            self.scratch, self.scratch_bytes = _DEFAULT_SCRATCH[0].data_ptr(), _DEFAULT_SCRATCH[1]


def set_default_scratch(tensor):
    """Attach `tensor` (zero-filled uint8 device memory of CONV_SCRATCH_BYTES, or None) to every ConvDesc created
    from now on that names no scratch buffer of its own (tests and micro-benchmarks; the engine passes its own)."""
    global _DEFAULT_SCRATCH
    _DEFAULT_SCRATCH = None if tensor is None else (tensor, tensor.numel() * tensor.element_size())


CONV_SCRATCH_BYTES = 4096 + 256 * 256 * 256 * 4      # VAMD_CONV_SCRATCH_BYTES


class RepackItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p), ("first_block", C.c_longlong),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("ksize", C.c_int32), ("ld_dgrad", C.c_int32),
                ("bias", C.c_void_p), ("bias_out", C.c_void_p), ("w_up4_dgrad", C.c_void_p), ("w_up4_fwd", C.c_void_p)]


class LinearItem(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("O", C.c_int32), ("first", C.c_int32)]


class LinearBwdItem(C.Structure):
    _fields_ = [("W", C.c_void_p), ("dy", C.c_void_p), ("dW", C.c_void_p), ("dbias", C.c_void_p), ("O", C.c_int32),
                ("pad_", C.c_int32)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first (python vision_amd/build_ext.py). "
        "vision_amd has no CPU or eager fallback.")

lib = C.CDLL(LIB_PATH)

_p, _i, _l, _f, _z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_d = C.POINTER(ConvDesc)

# name -> (restype, argtypes); mirrors include/vision_amd.h one to one
SIGNATURES = {
    "vamd_last_error": (C.c_char_p, []),
    "vamd_version": (_i, []),
    "vamd_launch_timing": (_i, [_i]),
    "vamd_launch_timing_read": (C.c_long, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vamd_conv2d_fwd": (_i, [_d, _p, _p, _p, _p, _p, _p, _p, _p]),
    "vamd_conv2d_fwd_stats": (_i, [_d, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "vamd_conv2d_fwd_act_supported": (_i, [_d]),
    "vamd_conv2d_fwd_act": (_i, [_d, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p]),
    "vamd_conv2d_stats_parts": (_i, [_d]),
    "vamd_conv2d_fwd_short_supported": (_i, [_d, _i]),
    "vamd_conv2d_fwd_short": (_i, [_d, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "vamd_fold_partials": (_i, [_p, _i, _i, _i, _p, _p]),
    "vamd_conv2d_dgrad": (_i, [_d, _p, _p, _p, _p, _p, _z, _p]),
    "vamd_conv2d_dgrad_workspace": (_z, [_d]),
    "vamd_conv2d_fwd_up4_supported": (_i, [_d]),
    "vamd_conv2d_fwd_up4_stats_parts": (_i, [_d]),
    "vamd_conv2d_fwd_up4": (_i, [_d, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "vamd_repack_weight_up4_fwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "vamd_conv2d_dgrad_up4_supported": (_i, [_d]),
    "vamd_conv2d_dgrad_up4": (_i, [_d, _p, _p, _p, _p, _p]),
    "vamd_repack_weight_up4": (_i, [_p, _p, _i, _i, _i, _p]),
    "vamd_conv2d_wgrad": (_i, [_d, _p, _p, _p, _p, _p, _z, _p]),
    "vamd_conv2d_wgrad_oihw": (_i, [_d, _p, _p, _p, _p, _p, _z, _p]),
    "vamd_conv2d_wgrad_workspace": (_z, [_d]),
    "vamd_conv2d_wgrad_oihw_split3": (_i, [_d, _p, _p, C.POINTER(C.c_void_p * 3), C.POINTER(C.c_void_p * 3), _p, _z, _p]),
    "vamd_conv2d_prenorm_supported": (_i, [_d]),
    "vamd_conv2d_fwd_prenorm": (_i, [_d, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p]),
    "vamd_conv2d_wgrad_prenorm_oihw": (_i, [_d, _p, _p, _p, _p, _f, _p, _p, _p, _z, _p]),
    "vamd_conv_splitk": (_i, [_i]),
    "vamd_conv2d_uses_mfma": (_i, [_d, _i]),
    "vamd_repack_weight": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "vamd_repack_weights": (_i, [_p, _i, C.c_longlong, _i, _p]),
    "vamd_unpack_wgrad": (_i, [_p, _p, _i, _i, _i, _p]),
    "vamd_stats_chunks": (_i, [_l, _i]),
    "vamd_stream_wait": (_i, [_p, _p]),
    "vamd_chan_stats": (_i, [_p, _p, _i, _l, _i, _i, _p]),
    "vamd_norm_bwd_stats": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _i, _l, _i, _i, _p]),
    "vamd_bn_finalize": (_i, [_p, _i, _i, _l, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p]),
    "vamd_gn_finalize": (_i, [_p, _i, _i, _i, _i, _l, _p, _p, _f, _p, _p, _p, _p, _p]),
    "vamd_gn_coef_noalloc": (_i, [_p, _i, _i, _i, _l, _p, _p, _f, _p, _p, _p, _p, _i, _p]),
    "vamd_scale_shift_act": (_i, [_p, _p, _p, _p, _f, _f, _p, _i, _l, _i, _i, _p]),
    "vamd_bn_bwd_finalize": (_i, [_p, _i, _i, _l, _p, _p, _p, _p, _p, _p, _i, _p]),
    "vamd_bn_bwd_stats_img": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _i, _l, _i, _i, _p]),
    "vamd_bn_bwd_finalize_img": (_i, [_p, _i, _i, _i, _l, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "vamd_gn_bwd_finalize": (_i, [_p, _i, _i, _i, _i, _l, _p, _p, _p, _p, _p, _p, _p, _p]),
    "vamd_norm_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p]),
    "vamd_sum_finalize": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "vamd_bmm_nt": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _p]),
    "vamd_bmm_tn": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    "vamd_sumpool2x2": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_pool2x2": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "vamd_unpool2x2": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "vamd_transpose": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_transpose_ld": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vamd_attention_fwd_supported": (_i, [_i, _i, _i]),
    "vamd_attention_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "vamd_softmax_rows": (_i, [_p, _p, _l, _i, _i, _i, _p]),
    "vamd_softmax_rows_bwd": (_i, [_p, _p, _p, _l, _i, _i, _f, _i, _p]),
    "vamd_add": (_i, [_p, _p, _p, _l, _i, _p]),
    "vamd_timeembed": (_i, [_p, _p, _p, _i, _i, _p]),
    "vamd_linear_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "vamd_linear_fwd_multi": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "vamd_linear_bwd_multi": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _p, _i, _p]),
    "vamd_linear_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "vamd_qsample": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_schedule": (_i, [_p, _p, _p, _p, _i, _p]),
    "vamd_l1_chunks": (_i, [_l]),
    "vamd_l1_loss_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_l1_loss_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_denoise_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_nchw_to_nhwc": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_adamw": (_i, [_p, _p, _p, _p, _l, _f, _p, _f, _f, _f, _f, _i, _f, _p]),
    "vamd_adam_l2": (_i, [_p, _p, _p, _p, _l, _f, _p, _f, _f, _f, _f, _i, _f, _p]),
    "vamd_adam_dev": (_i, [_p, _p, _p, _p, _l, _p, _p, _f, _f, _f, _f, _f, _i, _p]),
    "vamd_resize_ksize": (_i, [_i, _i]),
    "vamd_resize_coeffs": (_i, [_i, _i, _p, _p, _i]),
    "vamd_image_resize_u8": (_i, [_p, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p, _i, _i, _p, _p, _p]),
    "vamd_add_cast": (_i, [_p, _i, _p, _i, _p, _i, _l, _p]),
    "vamd_sigmoid_mse_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_sigmoid_mse_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vamd_vq_workspace": (_z, [_l]),
    "vamd_vq_forward": (_i, [_p, _p, _l, _i, _i, _i, _i, _f, _p, _p, _p, _p, _z, _p]),
    "vamd_vq_backward": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _f, _p, _p, _p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


def source_hash():
    """sha256 over the kernel sources and the header: profile summaries are stamped with it, so a reader can
    tell whether a committed counter pass still describes the kernels in the tree."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    for f in sorted(os.listdir(csrc)) + [os.path.join(_HERE, "..", "include", "vision_amd.h")]:
        with open(f if os.path.isabs(f) or f.startswith(_HERE) else os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(rc, what=""):
    if rc != 0:
        msg = lib.vamd_last_error().decode(errors="replace")
        raise VamdError(f"{what or 'libvision_amd'} failed (status {rc}): {msg}")


def call(name, *args):
    """Call an int-returning entry point and raise VamdError on a non-zero status."""
    check(getattr(lib, name)(*args), name)
