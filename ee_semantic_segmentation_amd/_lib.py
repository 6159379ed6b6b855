"""ctypes binding of libeeseg.so (the C ABI declared in include/eeseg.h).

The HIP library is the product: if it is missing or fails to load this module
raises - there is no CPU/PyTorch fallback anywhere in the package.
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads torch's bundled libamdhip64 FIRST so libeeseg binds to the same HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# EESEG_LIB: another build of the SAME ABI (same-box A/B of two builds); default = the in-tree library.  lib() refuses a
# library whose eeseg_version() differs from ABI_VERSION: the ctypes signatures below are written for exactly that ABI
LIB_PATH = os.environ.get("EESEG_LIB") or os.path.join(_HERE, "libeeseg.so")
ABI_VERSION = 106      # bumped with every signature / struct change of include/eeseg.h (csrc/api.hip returns the same number)

F32, BF16 = 0, 1


class EesegError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("residual", C.c_void_p),
                ("stats", C.c_void_p)] + \
               [(n, C.c_int) for n in ("N", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "R", "S",
                                       "smul", "off_h", "off_w", "tstep_h", "tstep_w", "sdiv",
                                       "ldy", "ldres", "relu", "dtype")] + \
               [("workspace", C.c_void_p), ("workspace_bytes", C.c_longlong), ("n_active", C.c_void_p),
                ("residual_mask", C.c_void_p), ("ld_residual_mask", C.c_int),
                ("n_taps", C.c_int), ("taps", C.c_void_p)]


class WgradArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p)] + \
               [(n, C.c_int) for n in ("N", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "R", "S",
                                       "stride", "pad", "dil", "dtype", "accumulate")] + \
               [("workspace", C.c_void_p), ("workspace_bytes", C.c_longlong), ("barrier_state", C.c_void_p)]


_vp, _i, _i64, _f, _d, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint64

# name -> (restype, argtypes); must list every symbol include/eeseg.h declares
SIGNATURES = {
    "eeseg_last_error": (C.c_char_p, []),
    "eeseg_version": (_i, []),
    "eeseg_set_option": (_i, [_i, _i]),
    "eeseg_get_option": (_i, [_i]),
    "eeseg_last_kernel": (_i, [_i]),
    "eeseg_set_ew_grid_cap": (_i, [_i]),
    "eeseg_set_wgrad_target_blocks": (_i, [_i]),
    "eeseg_set_wgrad_big": (_i, [_i]),
    "eeseg_set_wgrad_big_grid": (_i, [_i, _i]),
    "eeseg_set_wgrad_big_min_ktiles": (_i, [_i]),
    "eeseg_conv_stats_tiles": (_i, [_i, _i, _i]),
    "eeseg_conv_igemm": (_i, [C.POINTER(ConvArgs), _vp]),
    "eeseg_conv_workspace": (_i64, []),
    "eeseg_conv_wgrad": (_i, [C.POINTER(WgradArgs), _vp]),
    "eeseg_conv_wgrad_group": (_i, [C.POINTER(WgradArgs), _i, _vp]),
    "eeseg_set_wgrad_group": (_i, [_i]),
    "eeseg_wgrad_workspace": (_i64, []),
    "eeseg_pack_weight": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "eeseg_pack_weight_multi": (_i, [_vp, _i, _i, _vp]),
    "eeseg_pack_matrix": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "eeseg_im2col_nchw": (_i, [_vp, _vp] + [_i] * 12 + [_vp]),
    "eeseg_colreduce_workspace": (_i64, [_i64, _i]),
    "eeseg_bn_reduce_partials": (_i, [_vp, _i, _i, _vp, _vp, _i64, _vp]),
    "eeseg_bn_finalize": (_i, [_vp, _d, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _vp]),
    "eeseg_bn_reduce_finalize": (_i, [_vp, _i, _d, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _vp]),
    "eeseg_bn_eval_scale_shift": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _vp]),
    "eeseg_bn_apply": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i64, _i, _i, _i, _i, _vp]),
    "eeseg_bn_apply_relu_mask": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i64, _i, _i, _vp]),
    "eeseg_bn_finalize_apply": (_i, [_vp, _i, _vp, _d, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i64, _i, _i, _i,
                                     _vp]),
    "eeseg_bn_fwd_fused_ok": (_i, [_i64, _i, _i, _i]),
    "eeseg_bn_fwd_fused": (_i, [_vp, _i, _vp, _i, _d, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i64, _i, _i,
                                _i, _vp]),
    "eeseg_channel_stats": (_i, [_vp, _i, _i64, _i, _vp, _i, _vp, _i64, _vp]),
    "eeseg_bn_bwd_reduce": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i64, _i, _i, _vp, _vp, _i, _vp, _i64, _vp]),
    "eeseg_bn_bwd_apply": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _d, _vp, _i, _vp, _i, _i64, _i, _i,
                                _vp, _i, _vp]),
    "eeseg_bn_bwd_coop_ok": (_i, [_i64, _i, _i]),
    "eeseg_bn_bwd_coop_workspace": (_i64, []),
    "eeseg_bn_bwd_coop": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _d, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, _i, _i,
                               _vp, _i64, _vp, _vp]),
    "eeseg_scale_act_bwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i64, _i, _i, _i, _vp]),
    "eeseg_maxpool3x3s2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "eeseg_maxpool3x3s2_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "eeseg_sum_hw": (_i, [_vp, _i, _vp, _i, _i, _i, _f, _i, _vp, _i64, _vp]),
    "eeseg_broadcast_hw": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "eeseg_dropout": (_i, [_vp, _vp, _i64, _f, _u64, _vp, _i64, _i, _vp]),
    "eeseg_cast": (_i, [_vp, _i, _vp, _i, _i64, _vp]),
    "eeseg_add_inplace": (_i, [_vp, _vp, _i64, _i, _vp]),
    "eeseg_copy2d": (_i, [_vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "eeseg_colsum": (_i, [_vp, _i, _i64, _i, _vp, _i, _vp, _i64, _vp]),
    "eeseg_upsample_bilinear_nchw": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "eeseg_upsample_bilinear_nchw_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "eeseg_upsample_ce_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i64, _vp, _vp]),
    "eeseg_upsample_ce_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i64, _vp, _f, _vp, _vp, _vp]),
    "eeseg_argmax_confusion": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "eeseg_pil_bilinear_coeffs": (_i, [_i, _i, _vp, _vp, _i]),
    "eeseg_pil_nearest_index": (_i, [_i, _i, _vp]),
    "eeseg_preprocess_image_u8": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp,
                                       _vp, _vp, _vp]),
    "eeseg_preprocess_label_u8": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "eeseg_class_sums_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp]),
    "eeseg_class_sums_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp]),
    "eeseg_focal_map_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp]),
    "eeseg_focal_map_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp]),
    "eeseg_argmax_pair_hist": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "eeseg_entropy_gate_active": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "eeseg_argmax_exit": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "eeseg_exit_select": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "eeseg_gather_images": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "eeseg_ssim_labels": (_i, [_vp, _vp, _i, _i, _i, _d, _vp, _vp]),
    "eeseg_entropy_gate_workspace": (_i64, [_i, _i, _i]),
    "eeseg_entropy_gate": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i64, _vp]),
    "eeseg_lovasz_workspace": (_i64, [_i64, _i]),
    "eeseg_lovasz": (_i, [_vp, _vp, _i, _i, _i, _i64, _vp, _vp, _f, _vp, _u64, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    "eeseg_label_hist": (_i, [_vp, _i64, _i, _i64, _vp, _vp]),
    "eeseg_sgd_step": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _i, _vp]),
    "eeseg_comm_available": (_i, [C.POINTER(_i)]),
    "eeseg_comm_unique_id": (_i, [_vp]),
    "eeseg_comm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "eeseg_comm_destroy": (_i, [_vp]),
    "eeseg_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "eeseg_comm_check": (_i, [_vp]),
    "eeseg_comm_all_reduce": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "eeseg_comm_all_gather": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "eeseg_comm_broadcast": (_i, [_vp, _vp, _i64, _i, _vp]),
    "eeseg_comm_reduce_scatter": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
}

_lib = None


def lib():
    """The loaded library; raises EesegError when libeeseg.so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EesegError(
                f"{LIB_PATH} not found: the HIP extension is the product and there is no fallback. "
                "Build it with `python -m ee_semantic_segmentation_amd.build`.")
        l = C.CDLL(LIB_PATH)
        l.eeseg_version.restype = C.c_int
        if l.eeseg_version() != ABI_VERSION:
            raise EesegError(f"{LIB_PATH} implements ABI {l.eeseg_version()}, this package binds ABI {ABI_VERSION}: "
                             "rebuild it with `python -m ee_semantic_segmentation_amd.build`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise EesegError(f"{what} failed ({rc}): {lib().eeseg_last_error().decode()}")
