"""ctypes binding of libmi355unet3d.so (the C ABI declared in include/mi355_unet3d.h).

The product path has exactly one implementation: the HIP library built for gfx950 by `build.py`. If it is missing
or cannot be loaded this module raises -- there is no CPU or PyTorch fallback.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmi355unet3d.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)


class MiAct(Structure):
    _fields_ = [("p", c_void_p), ("n", c_int32), ("d", c_int32), ("h", c_int32), ("w", c_int32),
                ("c", c_int32), ("ld", c_int32), ("dtype", c_int32)]


class MiGnBwdFuse(Structure):
    """mi355_gn_bwd_fuse: the dgrad conv's epilogue also emits the first pass of the norm backward."""
    _fields_ = [("gx", c_void_p), ("gx_ld", c_int32), ("scale", c_void_p), ("shift", c_void_p), ("mean_rstd", c_void_p),
                ("groups", c_int32), ("act_slope", c_float), ("partials_out", c_void_p)]


class MiDiceOpts(Structure):
    """mi355_dice_opts: the monai DiceLoss options beyond the shipped configuration."""
    _fields_ = [("activation", c_int32), ("target_kind", c_int32), ("batch", c_int32), ("squared_pred", c_int32),
                ("include_background", c_int32), ("jaccard", c_int32), ("reduction", c_int32),
                ("smooth_nr", c_float), ("smooth_dr", c_float), ("class_weight", c_void_p)]


class MiConvDesc(Structure):
    _fields_ = [("kd", c_int32), ("stride", c_int32), ("pad", c_int32), ("in_mode", c_int32),
                ("act_slope", c_float),
                ("in_scale", c_void_p), ("in_shift", c_void_p), ("bias", c_void_p),
                ("residual", c_void_p), ("residual_ld", c_int32),
                ("out_chscale", c_void_p),
                ("off_z", c_int32), ("off_y", c_int32), ("off_x", c_int32),
                ("out_d", c_int32), ("out_h", c_int32), ("out_w", c_int32),
                ("in_slope", c_void_p), ("out_mode", c_int32), ("precision", c_int32), ("wformat", c_int32),
                ("moments_out", c_void_p), ("gn_bwd", POINTER(MiGnBwdFuse))]


IN_PLAIN, IN_AFFINE_ACT, IN_ZERO_INSERT, IN_S2D = 0, 1, 2, 3
OUT_PLAIN, OUT_D2S = 0, 1
W_PACKED, W_OIDHW4, W_PACKED_F32_NARROW = 0, 1, 2
PREC_F32, PREC_BF16X3, PREC_BF16X6, PREC_BF16, PREC_F16 = 0, 1, 2, 3, 4
ACT_F32, ACT_BF16, ACT_F16 = 0, 1, 2      # mi355_act.dtype
PRECISIONS = {"fp32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16x6": PREC_BF16X6, "bf16": PREC_BF16, "fp16": PREC_F16}

STATUS = {0: "ok", -1: "invalid argument (shape/alignment/null)", -2: "unsupported combination",
          -3: "kernel launch failed", -4: "workspace too small"}

# name -> (restype, argtypes); every symbol of include/mi355_unet3d.h
SIGNATURES = {
    "mi355_packed_weight_elems": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "mi355_pack_conv_weight": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mi355_pack_weights_batch": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "mi355_packed_weight_bytes_bf16": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "mi355_pack_conv_weight_bf16": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mi355_conv3d_uses_bf16": (ctypes.c_int, [POINTER(MiConvDesc)]),
    "mi355_conv3d_fwd": (ctypes.c_int, [POINTER(MiAct), c_void_p, POINTER(MiAct), POINTER(MiConvDesc), c_void_p]),
    "mi355_conv3d_stats_blocks": (c_int32, [POINTER(MiAct), POINTER(MiAct), POINTER(MiConvDesc)]),
    "mi355_conv3d_fwd_config": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), POINTER(MiConvDesc), c_char_p, c_size_t]),
    "mi355_conv3d_wgrad_workspace": (c_size_t, [POINTER(MiAct), POINTER(MiAct), POINTER(MiConvDesc)]),
    "mi355_conv3d_wgrad": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), c_void_p, POINTER(MiConvDesc), c_void_p, c_size_t, c_void_p]),
    "mi355_gn_workspace": (c_size_t, [POINTER(MiAct)]),
    "mi355_gn_stats": (ctypes.c_int, [POINTER(MiAct), c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mi355_gn_moments_blocks": (c_int32, [POINTER(MiAct)]),
    "mi355_gn_moments": (ctypes.c_int, [POINTER(MiAct), c_void_p, c_void_p]),
    "mi355_gn_records_reduce": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "mi355_gn_finalize": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "mi355_gn_act_bwd_fused": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), POINTER(MiAct), c_void_p, c_int32, c_int32, c_float,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t,
                                              c_void_p]),
    "mi355_gn_act_bwd": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), POINTER(MiAct), c_void_p, c_int32, c_int32, c_float,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mi355_upsample2x_fwd": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), c_int32, c_int32, c_int32, c_void_p]),
    "mi355_upsample2x_bwd": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), c_int32, c_int32, c_int32, c_void_p]),
    "mi355_ncdhw_to_ndhwc": (ctypes.c_int, [c_void_p, POINTER(MiAct), c_void_p]),
    "mi355_ndhwc_to_ncdhw": (ctypes.c_int, [POINTER(MiAct), c_void_p, c_void_p]),
    "mi355_add": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), POINTER(MiAct), c_void_p]),
    "mi355_chscale": (ctypes.c_int, [POINTER(MiAct), c_void_p, POINTER(MiAct), c_void_p]),
    "mi355_cast": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), c_void_p]),
    "mi355_proj_fwd": (ctypes.c_int, [POINTER(MiAct), c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "mi355_proj_workspace": (c_size_t, [POINTER(MiAct), c_int32]),
    "mi355_proj_bwd": (ctypes.c_int, [POINTER(MiAct), c_void_p, c_void_p, c_float, c_void_p, c_void_p, POINTER(MiAct), c_void_p, c_void_p, c_int32,
                                      c_void_p, c_size_t, c_void_p]),
    "mi355_sw_accumulate": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                           c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mi355_sw_gather": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                       c_void_p, c_void_p]),
    "mi355_sw_accumulate_batch": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                 c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "mi355_sw_normalize": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "mi355_postprocess": (ctypes.c_int, [c_void_p, c_int32, c_int64, c_int32, c_float, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "mi355_one_hot": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "mi355_zscore_workspace": (c_size_t, [c_int32]),
    "mi355_zscore": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_size_t, c_void_p]),
    "mi355_resample_affine": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                             POINTER(ctypes.c_float), c_int32, c_int32, c_void_p]),
    "mi355_wino_weight_elems": (c_size_t, [c_int32, c_int32]),
    "mi355_wino_pack_weight": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mi355_conv3d_wino_fwd": (ctypes.c_int, [POINTER(MiAct), c_void_p, POINTER(MiAct), POINTER(MiConvDesc), c_void_p]),
    "mi355_conv3d_wino_supported": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), POINTER(MiConvDesc)]),
    "mi355_conv3d_wino_stats_blocks": (c_int32, [POINTER(MiAct)]),
    "mi355_conv3d_c4_bwd_supported": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), POINTER(MiConvDesc)]),
    "mi355_conv3d_c4_bwd_blocks": (c_int32, [POINTER(MiAct)]),
    "mi355_conv3d_c4_bwd_workspace": (c_size_t, [POINTER(MiAct), POINTER(MiAct), POINTER(MiConvDesc)]),
    "mi355_conv3d_c4_bwd": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), c_void_p, c_void_p, POINTER(MiConvDesc), c_void_p, c_int32, c_void_p,
                                           c_void_p, c_size_t, c_void_p]),
    "mi355_gn_bwd_params": (ctypes.c_int, [POINTER(MiAct), c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t,
                                           c_void_p]),
    "mi355_wino3d_weight_elems": (c_size_t, [c_int32, c_int32]),
    "mi355_wino3d_pack_weight": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mi355_conv3d_wino3d_fwd": (ctypes.c_int, [POINTER(MiAct), c_void_p, POINTER(MiAct), POINTER(MiConvDesc), c_void_p]),
    "mi355_conv3d_wino3d_supported": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), POINTER(MiConvDesc)]),
    "mi355_conv3d_wgrad_wino_workspace": (c_size_t, [POINTER(MiAct), POINTER(MiAct), POINTER(MiConvDesc)]),
    "mi355_conv3d_wgrad_wino": (ctypes.c_int, [POINTER(MiAct), POINTER(MiAct), c_void_p, POINTER(MiConvDesc), c_void_p, c_size_t, c_void_p]),
    "mi355_dice_workspace": (c_size_t, [c_int32, c_int32, c_int64]),
    "mi355_dice_fwd_bwd": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_float, c_float, c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "mi355_dice_ex_forward": (ctypes.c_int, [POINTER(MiDiceOpts), c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_size_t,
                                             c_void_p]),
    "mi355_dice_ex_backward": (ctypes.c_int, [POINTER(MiDiceOpts), c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_int32, c_void_p,
                                              c_void_p, c_void_p]),
    "mi355_ce_workspace": (c_size_t, [c_int64]),
    "mi355_ce_fwd_bwd": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int32, c_float, c_void_p, c_int32,
                                        c_void_p, c_int32, c_float, c_void_p, c_size_t, c_void_p]),
    "mi355_adam_step": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_double, c_double,
                                       c_double, c_int32, c_float, c_void_p]),
    "mi355_version": (c_char_p, []),
}


def bind(cdll):
    """Attach restype/argtypes for every declared symbol; raises AttributeError if one is missing."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


_cached = None


def load_library(path=None):
    """Load the HIP library. torch must be imported first so that its bundled libamdhip64.so.7 is the HIP runtime
    both sides share (same SONAME as the toolkit's; streams and pointers are then interchangeable)."""
    global _cached
    if path is None and _cached is not None:
        return _cached
    import torch  # noqa: F401  (loads torch's HIP runtime before ours resolves libamdhip64.so.7)
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the MI355X HIP library has not been built. Run `python __graft_entry__.py build` "
            "(or `python 3dunetcnn_amd/build.py`). There is no CPU fallback for this package.")
    lib = bind(ctypes.CDLL(p))
    if path is None:
        _cached = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: mi355 status {rc} ({STATUS.get(rc, 'unknown')})")
