"""ctypes binding of libsfh_amd.so (C ABI: include/sfh_amd.h).

There is deliberately no fallback: if the library is missing or a symbol cannot be
resolved, importing the hot path fails with an explicit error.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFH_AMD_LIB overrides the library file (used only to load the diagnostic build for profiling)
LIB_PATH = os.environ.get("SFH_AMD_LIB") or os.path.join(_HERE, "libsfh_amd.so")

TILE_8x32, TILE_16x16, TILE_32x8, TILE_8x16, TILE_16x8 = 0, 1, 2, 3, 4
OUT_NHWC, OUT_UPSCATTER2 = 0, 1
FMT_F32, FMT_S3, FMT_H2, FMT_FH2 = 0, 1, 2, 3
H2_ACT_EXP = 2   # SFH_H2_ACT_EXP: the default exponent of an H2 tensor
H2_LIMIT_BITS = 0x477FE000   # bit pattern of 65504.f: sfh_conv_desc.h2_range words above it mean saturation

_p = C.c_void_p
_i = C.c_int32


class ConvDesc(C.Structure):
    """Mirror of ``struct sfh_conv_desc`` (include/sfh_amd.h)."""
    _fields_ = [
        ("src0", _p), ("c0", _i), ("cs0", _i), ("h0", _i), ("w0", _i), ("pool0", _i),
        ("src1", _p), ("c1", _i), ("cs1", _i), ("h1", _i), ("w1", _i), ("pad_top1", _i), ("pad_left1", _i),
        ("batch", _i), ("H", _i), ("W", _i), ("ksize", _i), ("stride", _i), ("tile", _i),
        ("wpacked", _p), ("scale", _p), ("shift", _p), ("cout", _i), ("relu", _i),
        ("residual", _p),
        ("dst", _p), ("dst_cs", _i), ("out_mode", _i),
        ("src_fmt", _i), ("dst_fmt", _i), ("dst_pool", _p), ("pool_cs", _i),
        ("residual_f32", _i), ("shift_border", _p), ("up_dst_h", _i), ("up_dst_w", _i),
        ("reverse_tiles", _i),
        ("head_w", _p), ("head_b", _p), ("head_nc", _i), ("head_skip_dst", _i),
        ("head_logits", _p), ("head_stn", _p), ("head_frame", _p),
        ("h2_overflow", _p), ("h2_exp_src", _i), ("h2_exp_dst", _i), ("h2_exp_res", _i), ("h2_range", _p),
        ("wg_couts", _i), ("split_arith", _i), ("ksplit", _i), ("ksplit_stride", C.c_int64), ("acc_init", _p),
        ("stats_partial", _p), ("stats_rows", _i),
        ("bwd_z", _p), ("bwd_mi", _p), ("bwd_gamma", _p), ("bwd_beta", _p),
        ("up_wpacked", _p), ("up_scale", _p),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        # H2 tensors of a launch carry v * 2^e; callers that do not manage exponents (the training kernels) get the
        # conventional default for every tensor
        if "h2_exp_src" not in kw:
            self.h2_exp_src = self.h2_exp_dst = self.h2_exp_res = H2_ACT_EXP


# name -> (restype, argtypes); every symbol declared in include/sfh_amd.h
SIGNATURES = {
    "sfh_last_error": (C.c_char_p, []),
    "sfh_version": (C.c_int, []),
    "sfh_conv_fwd": (C.c_int, [C.POINTER(ConvDesc), _p]),
    "sfh_conv_s3_fwd": (C.c_int, [C.POINTER(ConvDesc), _p]),
    "sfh_packed_s3_weight_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sfh_pack_s3_weights": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_f32_to_s3": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, _p]),
    "sfh_s3_to_f32": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, _p]),
    "sfh_packed_h2_weight_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sfh_pack_h2_weights": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_f32_to_h2": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_h2_to_f32": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_conv3x3_c4_fwd": (C.c_int, [C.POINTER(ConvDesc), _p]),
    "sfh_pack_c4_weights": (C.c_int, [_p, _p, C.c_int, C.c_int, _p]),
    "sfh_frame_to_h2": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_packed_c4h2_weight_bytes": (C.c_int64, [C.c_int]),
    "sfh_pack_c4h2_weights": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_conv3x3_c4h2_fwd": (C.c_int, [C.POINTER(ConvDesc), _p]),
    "sfh_conv_small_fwd": (C.c_int, [C.POINTER(ConvDesc), _p]),
    "sfh_conv_upfused_fwd": (C.c_int, [C.POINTER(ConvDesc), _p]),
    "sfh_packed_weight_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sfh_pack_conv_weights": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_space_to_depth2": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_fold_bn": (C.c_int, [_p, _p, _p, _p, _p, C.c_float, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_u8hwc_to_f32nchw": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_u8hwc_area2_to_f32nchw": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_u8hwc_areak_to_f32nchw": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_u8hwc_areaxy_to_f32nchw": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_resize_area_tab": (C.c_int, [C.c_int, C.c_int, _p, _p, _p, C.c_int]),
    "sfh_u8hwc_area_to_f32nchw": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p, _p, _p]),
    "sfh_nchw_to_nhwc": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_nhwc_to_nchw": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_resize_nchw": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_upsample2x_bilinear_nhwc": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_outconv_fwd": (C.c_int, [_p, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p,
                                  _p, C.c_int, _p, C.c_int, _p]),
    "sfh_selftest_warp_arith": (C.c_int, [_p, _p]),
    "sfh_homography_warp_fwd": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_float, _p, _p, _p]),
    "sfh_poi_project_fwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p, _p]),
    "sfh_ce_workspace_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "sfh_consistency_ce_fwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         _p, _p, _p]),
    "sfh_mask_format_fwd": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, _p, _p, _p]),
    "sfh_bn_stats": (C.c_int, [_p, C.c_int64, C.c_int, _p, _p]),
    "sfh_bn_stats_partials": (C.c_int, [_p, C.c_int, C.c_int, _p, _p]),
    "sfh_bn_finalize": (C.c_int, [_p, C.c_int64, C.c_int, C.c_float, C.c_float, _p, _p, _p, _p, _p]),
    "sfh_bn_finalize_partials": (C.c_int, [_p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_float, _p, _p, _p, _p, _p]),
    "sfh_bn_apply": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, C.c_int64, C.c_int, _p, _p, C.c_int, C.c_int, _p, _p]),
    "sfh_bn_bwd_reduce": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int64, C.c_int, _p, _p]),
    "sfh_bn_bwd_apply": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, C.c_int, C.c_int64, C.c_int, _p, _p, _p, C.c_int, C.c_int,
                                   _p, _p, _p]),
    "sfh_colsum": (C.c_int, [_p, C.c_int64, C.c_int, C.c_int, _p, _p]),
    "sfh_s2d_split_colsum": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.c_int, _p, C.c_int, _p, C.c_int, _p, _p]),
    "sfh_maxpool2_fwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_bn_apply_pool": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, C.c_int, _p, _p]),
    "sfh_pool2_bwd_bn_reduce": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_maxpool2_bwd": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_slice_add": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_zero_stuff2": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_conv_wgrad_c4_bn": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, C.c_int, _p]),
    "sfh_conv_wgrad": (C.c_int, [_p, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, _p, C.c_int, C.c_int, _p]),
    "sfh_conv_wgrad_s3": (C.c_int, [_p, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_outconv_bwd": (C.c_int, [_p, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p]),
    "sfh_outconv_bwd_bn": (C.c_int, [_p, _p, _p, _p, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p]),
    "sfh_maxpool3x3s2_bwd": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_avgpool_linear_bwd": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p]),
    "sfh_stem_bwd_data": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "sfh_homography_warp_bwd_theta": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                _p, _p, _p]),
    "sfh_poi_project_bwd_theta": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_train_losses": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                   C.c_int, C.c_float, C.c_int, _p, _p, _p, _p]),
    "sfh_sgd_step": (C.c_int, [_p, _p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _p]),
    "sfh_adam_step": (C.c_int, [_p, _p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int, _p]),
    "sfh_uv_loss": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _p, _p, _p]),
    "sfh_multi_copy_dscale": (C.c_int, [_p, _p, C.c_int, _p, _p]),
    "sfh_grad_scale": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_multi_absminmax": (C.c_int, [_p, C.c_int, _p, _p]),
    "sfh_vec_op": (C.c_int, [C.c_int, _p, _p, C.c_int64, C.c_int, C.c_float, _p, _p]),
    "sfh_copy2d_words": (C.c_int, [_p, C.c_int64, _p, C.c_int64, C.c_int, C.c_int64, _p]),
    "sfh_fill_words": (C.c_int, [_p, C.c_int64, C.c_uint32, _p]),
    "sfh_rows_differ": (C.c_int, [_p, C.c_int64, C.c_int, _p, _p]),
    "sfh_stn_input_assemble": (C.c_int, [_p, C.c_int, _p, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "sfh_reproj_loss": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_float, _p, _p, _p]),
    "sfh_rmsprop_step": (C.c_int, [_p, _p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_float, _p]),
    "sfh_multi_copy": (C.c_int, [_p, _p, C.c_int, C.c_float, _p]),
    "sfh_upsample2x_bilinear_nhwc_bwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_resize_nearest_nchw_bwd": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_compose_up_weights": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p, _p]),
    "sfh_splitk_finish": (C.c_int, [_p, C.c_int, C.c_int64, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                    _p, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_stem7x7_fwd": (C.c_int, [C.POINTER(ConvDesc), _p]),
    "sfh_packed_stem_weight_bytes": (C.c_int64, []),
    "sfh_pack_stem_weights": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_maxpool3x3s2_fwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "sfh_maxpool3x3s2_split_fwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_avgpool_linear_fwd": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "sfh_probe_mfma_f16": (C.c_int, [C.c_int, C.c_int, _p, _p, _p]),
    "sfh_warp_consistency_workspace_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "sfh_warp_consistency_fwd": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _p, C.c_int,
                                           C.c_int, C.c_int, _p, _p, _p, _p]),
}

_lib = None


class SfhError(RuntimeError):
    pass


def load():
    """Load libsfh_amd.so and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SfhError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU/PyTorch fallback for this path."
        )
    # PyTorch first: it ships its own HIP runtime (torch/lib/libamdhip64.so) and owns device memory and streams.
    # Loading libsfh_amd.so before torch would bring /opt/rocm's copy of the runtime into the process as well,
    # and the second copy then sees no device ("no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().sfh_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise SfhError(f"{what}: error {rc}: {msg}")
