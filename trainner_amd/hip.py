"""ctypes binding of libtrainner_hip.so (the C ABI declared in include/trainner_hip.h).

The product path has NO CPU or eager-PyTorch fallback: if the library is missing, or a call is
made without a HIP device, this module raises.  torch is used only to own device memory and
streams (`tensor.data_ptr()`, `torch.cuda.current_stream()`).
"""
import ctypes as C
import os

import torch

from . import build as _build

c_f = C.c_float
c_i = C.c_int32
c_l = C.c_int64
c_p = C.c_void_p

ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2
MMA_F32, MMA_BF16, MMA_BF16X3 = 0, 1, 2
CONV_3x3, CONV_3x3_UP2, CONV_4x4_S2, DGRAD_4x4_S2, CONV_1x1, CONV_3x3_C4, CONV_7x7_C4 = 0, 1, 2, 3, 4, 5, 6
PACK_FWD, PACK_DGRAD_3x3, PACK_FWD_S2D, PACK_DGRAD_S2, PACK_COL_FWD, PACK_COL_DGRAD3 = 0, 1, 2, 3, 4, 5
PACK_C4_FWD, PACK_C4_DGRAD3 = 6, 7
PACK_DENSE_DGRAD = 16      # host-side tag of tnr_pack_dense_dgrad slabs (consumed like PACK_FWD by conv_tile)


class CView(C.Structure):
    _fields_ = [("ptr", c_p), ("ctot", c_i), ("coff", c_i)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", CView), ("N", c_i), ("H", c_i), ("W", c_i), ("Cin", c_i),
        ("wp", c_p), ("KinP", c_i), ("KoutP", c_i),
        ("y", CView), ("Ho", c_i), ("Wo", c_i), ("Cout", c_i),
        ("mode", c_i), ("bias", c_p), ("act", c_i), ("slope", c_f), ("alpha", c_f),
        ("r1", CView), ("r1_ch", c_i), ("beta1", c_f),
        ("r2", CView), ("alpha2", c_f),
        ("m", CView), ("m_lo", c_i), ("m_hi", c_i), ("m_slope", c_f),
        ("ws", c_p), ("ws_bytes", c_l), ("mma", c_i), ("pad_mode", c_i),
        ("noise_sigma", c_f), ("noise_pos", c_i), ("noise_key0", C.c_uint32), ("noise_key1", C.c_uint32), ("noise_pix0", C.c_uint32),
        ("wq", c_p), ("wq_bytes", c_l), ("wq_form", c_i), ("shuffle", c_i),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", CView), ("N", c_i), ("H", c_i), ("W", c_i), ("Cin", c_i),
        ("g", CView), ("Ho", c_i), ("Wo", c_i), ("Cout", c_i),
        ("mode", c_i),
        ("dw", c_p), ("cin_total", c_i), ("cin_begin", c_i),
        ("db", c_p), ("alpha", c_f), ("beta", c_f),
        ("ws", c_p), ("ws_bytes", c_l), ("mma", c_i), ("pad_mode", c_i),
        ("dw2", c_p), ("cout_split", c_i), ("cin_total2", c_i), ("db2", c_p),
    ]


class SweepPackItem(C.Structure):      # tnr_sweep_pack_item
    _fields_ = [("units", c_i), ("reserved", c_i), ("opaque", C.c_uint64 * 15)]


class PackItem(C.Structure):
    _fields_ = [("w", c_p), ("wp", c_p), ("Cout", c_i), ("Cin", c_i), ("kh", c_i), ("kw", c_i),
                ("kind", c_i), ("KoutP", c_i), ("KinP", c_i), ("n_out", c_l)]


class DensePackItem(C.Structure):
    _fields_ = [("w", c_p * 5), ("wp", c_p), ("nf", c_i), ("gc", c_i), ("t", c_i), ("KoutP", c_i), ("KinP", c_i),
                ("scale5", c_f), ("n_out", c_l)]


_SIGS = {
    "tnr_last_error": (C.c_char_p, []),
    "tnr_version": (c_i, []),
    "tnr_pack_dims": (c_i, [c_i, c_i, c_i, c_i, c_i, C.POINTER(c_i), C.POINTER(c_i), C.POINTER(c_l)]),
    "tnr_pack_weights": (c_i, [c_p, c_i, c_l, c_p]),
    "tnr_pack_dense_dims": (c_i, [c_i, c_i, c_i, C.POINTER(c_i), C.POINTER(c_i), C.POINTER(c_l)]),
    "tnr_pack_dense_dgrad": (c_i, [c_p, c_i, c_l, c_p]),
    "tnr_conv_forward": (c_i, [C.POINTER(ConvDesc), c_p]),
    "tnr_conv_workspace_bytes": (c_l, [C.POINTER(ConvDesc)]),
    "tnr_conv_wq_bytes": (c_l, [C.POINTER(ConvDesc)]),
    "tnr_conv_wq_pack": (c_i, [C.POINTER(ConvDesc), c_p, c_l, c_p]),
    "tnr_conv_wino_bytes": (c_l, [C.POINTER(ConvDesc)]),
    "tnr_conv_wino_pack": (c_i, [C.POINTER(ConvDesc), c_p, c_l, c_p]),
    "tnr_gauss_mult": (c_i, [CView, CView, c_l, c_i, c_f, C.c_uint32, C.c_uint32, C.c_uint32, c_p]),
    "tnr_im2col": (c_i, [CView, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "tnr_conv_chain_workspace_bytes": (c_l, [C.POINTER(ConvDesc)]),
    "tnr_conv_chain": (c_i, [C.POINTER(ConvDesc), C.POINTER(C.c_int32), c_i, c_p, c_l, C.c_uint32, c_p]),
    "tnr_conv_sweep_image_bytes": (c_l, [C.POINTER(ConvDesc), c_i]),
    "tnr_conv_sweep_pack_item": (c_i, [C.POINTER(ConvDesc), c_i, c_p, c_l, C.POINTER(SweepPackItem)]),
    "tnr_conv_sweep_pack_batch": (c_i, [c_p, c_i, c_i, c_p]),
    "tnr_conv_sweep_pack": (c_i, [C.POINTER(ConvDesc), c_i, c_p, c_l, c_p]),
    "tnr_conv_sweep": (c_i, [C.POINTER(ConvDesc), c_i, c_p, c_p, c_l, C.c_uint32, c_p]),
    "tnr_conv_thin_pack_floats": (c_l, [c_i]),
    "tnr_conv_thin_pack": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "tnr_conv_thin": (c_i, [CView, c_i, c_i, c_i, c_i, c_p, CView, c_i, c_p, c_f, c_p]),
    "tnr_wgrad_workspace_bytes": (c_l, [C.POINTER(WgradDesc)]),
    "tnr_conv_thin7_pack_floats": (c_l, [c_i]),
    "tnr_conv_thin7_pack": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "tnr_conv_thin7": (c_i, [CView, c_i, c_i, c_i, c_i, c_p, CView, c_i, c_i, c_i, c_i, c_i, c_p, c_f, c_p]),
    "tnr_wgrad_thin7_workspace_bytes": (c_l, [c_i, c_i, c_i]),
    "tnr_wgrad_thin7": (c_i, [CView, c_i, c_i, c_i, c_i, CView, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_p, c_l, c_p]),
    "tnr_wgrad_thin_workspace_bytes": (c_l, [c_i, c_i, c_i]),
    "tnr_wgrad_thin": (c_i, [CView, CView, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_p, c_l, c_p]),
    "tnr_conv_wgrad": (c_i, [C.POINTER(WgradDesc), c_p]),
    "tnr_conv_wgrad_group": (c_i, [C.POINTER(WgradDesc), c_i, c_p]),
    "tnr_nchw_to_nhwc": (c_i, [c_p, c_i, c_i, c_i, c_i, CView, c_i, c_p, c_p, c_p]),
    "tnr_nhwc_to_nchw": (c_i, [CView, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p]),
    "tnr_upsample2x_bwd": (c_i, [CView, CView, c_i, c_i, c_i, c_i, CView, c_f, c_p]),
    "tnr_depth_to_space": (c_i, [CView, CView, c_i, c_i, c_i, c_i, c_p]),
    "tnr_space_to_depth_bwd": (c_i, [CView, CView, c_i, c_i, c_i, c_i, CView, c_f, c_p]),
    "tnr_maxpool2_fwd": (c_i, [CView, CView, c_i, c_i, c_i, c_i, c_p]),
    "tnr_maxpool2_bwd": (c_i, [CView, CView, CView, c_i, c_i, c_i, c_i, c_p]),
    "tnr_gconv_fwd": (c_i, [CView, c_i, c_i, c_i, c_i, c_p, c_p, CView, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "tnr_gconv_dgrad": (c_i, [CView, c_i, c_i, c_i, c_i, c_p, CView, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "tnr_gconv_wgrad_workspace_bytes": (c_l, [c_i, c_i, c_i]),
    "tnr_gconv_wgrad": (c_i, [CView, c_i, c_i, c_i, c_i, CView, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_p, c_l, c_p]),
    "tnr_bias_grad": (c_i, [CView, c_l, c_i, c_p, c_f, c_f, c_p, c_l, c_p]),
    "tnr_pad2d": (c_i, [CView, CView, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "tnr_unpad2d": (c_i, [CView, CView, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "tnr_window2d": (c_i, [CView, c_i, c_i, CView, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "tnr_tanh_fwd": (c_i, [c_p, c_p, c_l, c_p]),
    "tnr_tanh_bwd": (c_i, [c_p, c_p, c_p, c_l, c_p]),
    "tnr_gan_loss": (c_i, [c_p, c_l, c_i, c_f, c_p, c_p, c_p]),
    "tnr_dp_unique_id": (c_i, [c_p]),
    "tnr_dp_init": (c_i, [c_p, c_i, c_i, C.POINTER(c_p)]),
    "tnr_dp_allreduce_bucket": (c_i, [c_p, c_p, c_l, c_i, c_p]),
    "tnr_dp_broadcast": (c_i, [c_p, c_p, c_l, c_i, c_p]),
    "tnr_dp_comm_count": (c_i, [c_p, C.POINTER(c_i)]),
    "tnr_dp_finalize": (c_i, [c_p]),
    "tnr_filter2d": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "tnr_resize": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "tnr_noise_gaussian": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, C.c_uint64, c_i, c_p]),
    "tnr_noise_poisson": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, C.c_uint64, c_i, c_p]),
    "tnr_jpeg_workspace_bytes": (c_l, [c_i, c_i, c_i]),
    "tnr_jpeg_sim": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_l, c_p]),
    "tnr_feed_u8_to_tensor": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_f, c_i, c_p]),
    "tnr_tensor2np_u8": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "tnr_metrics_workspace_bytes": (c_l, [c_i]),
    "tnr_psnr_ssim_u8": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p]),
    "tnr_bilinear2x_fwd": (c_i, [CView, CView, c_i, c_i, c_i, c_i, c_p]),
    "tnr_bilinear2x_bwd": (c_i, [CView, CView, CView, CView, c_f, c_i, c_i, c_i, c_i, c_p]),
    "tnr_add2": (c_i, [CView, CView, CView, c_l, c_i, c_p]),
    "tnr_mask_copy": (c_i, [CView, CView, CView, c_l, c_i, c_f, c_p]),
    "tnr_axpby": (c_i, [CView, CView, c_l, c_i, c_f, c_f, c_p]),
    "tnr_mask_mul": (c_i, [CView, CView, c_l, c_i, c_f, c_p]),
    "tnr_fill": (c_i, [c_p, c_l, c_f, c_p]),
    "tnr_bn_workspace_bytes": (c_l, [c_i]),
    "tnr_bn_train_fwd": (c_i, [CView, CView, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_i, c_f, c_p, c_p]),
    "tnr_bn_train_fwd_stats": (c_i, [CView, CView, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_i, c_f, c_p, c_p]),
    "tnr_bn_replay_running": (c_i, [c_p, c_p, c_p, c_p, c_i, c_f, c_p]),
    "tnr_bn_train_bwd": (c_i, [CView, CView, CView, CView, c_l, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_f, c_p, c_p]),
    "tnr_bn_train_bwd_z": (c_i, [CView, CView, CView, c_l, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_f, c_p, c_p]),
    "tnr_instnorm_workspace_bytes": (c_l, [c_i, c_i]),
    "tnr_instnorm_fwd": (c_i, [CView, CView, c_i, c_l, c_i, c_f, c_p, c_p, c_i, c_f, c_p, c_p]),
    "tnr_instnorm_bwd": (c_i, [CView, CView, CView, CView, c_i, c_l, c_i, c_p, c_p, c_f, c_p, c_p]),
    "tnr_linear_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "tnr_linear_bwd": (c_i, [c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p]),
    "tnr_reduce_workspace_bytes": (c_l, []),
    "tnr_l1_mean_fwd": (c_i, [c_p, c_p, c_l, c_f, c_p, c_p, c_p]),
    "tnr_l1_mean_bwd": (c_i, [c_p, c_p, c_l, c_f, c_p, c_p, c_i, c_p]),
    "tnr_ragan_phase_a": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p]),
    "tnr_ragan_phase_b": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    "tnr_ragan_phase_c": (c_i, [c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p]),
    "tnr_scale_by": (c_i, [c_p, c_p, c_l, c_p, c_p]),
    "tnr_sumsq": (c_i, [c_p, c_l, c_p, c_p, c_p]),
    "tnr_clip_by_norm": (c_i, [c_p, c_l, c_p, c_f, c_p]),
    "tnr_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_p]),
    "tnr_adam_step_guarded": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p]),
    "tnr_set_fault_word": (c_i, [c_p]),
}

EXPORTS = tuple(_SIGS)
_lib = None


class HipEngineError(RuntimeError):
    pass


def library_path():
    """In-tree build by default; TNR_HIP_LIB points at another build of the same C ABI (kernel experiments)."""
    return os.environ.get("TNR_HIP_LIB") or _build.LIB


ABI_VERSION = 3          # include/trainner_hip.h TNR_ABI_VERSION: the descriptor layouts below are those of this version


def load(build_if_missing=False):
    """dlopen the in-tree library and type every export.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        if build_if_missing:
            _build.build()
        else:
            raise HipEngineError(
                "libtrainner_hip.so not found at %s -- run `python -m trainner_amd.build` "
                "(the engine has no CPU / eager fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.tnr_version() != ABI_VERSION:
        raise HipEngineError("%s speaks ABI version %d, this binding expects %d (include/trainner_hip.h TNR_ABI_VERSION): rebuild with "
                             "`python -m trainner_amd.build --force`" % (path, lib.tnr_version(), ABI_VERSION))
    _lib = lib
    return lib


def require_device(t=None):
    if not torch.cuda.is_available():
        raise HipEngineError("no HIP device visible: the trainner_amd engine only runs on MI355X (gfx950)")
    if t is not None and not t.is_cuda:
        raise HipEngineError("tensor is not on a HIP device")


def engine_device(index=0):
    require_device()
    return torch.device("cuda", index)


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what=""):
    if rc != 0:
        msg = load().tnr_last_error()
        raise HipEngineError("%s failed (%d): %s" % (what or "tnr call", rc, msg.decode() if msg else "?"))


NULLVIEW = CView(None, 0, 0)


def ptr(t):
    return None if t is None else t.data_ptr()
