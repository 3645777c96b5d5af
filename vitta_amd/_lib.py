"""ctypes binding of libvitta_hip.so (the C ABI declared in include/vitta_hip.h).

The product path never falls back: if the shared library is missing or a symbol
cannot be resolved, `lib()` raises.  Building happens in `vitta_amd.build`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvitta_hip.so")

LAYOUT_NCHW = 0
LAYOUT_NHWC = 1
REG_TYPES = {"l1_loss": 0, "mse_loss": 1, "kld": 2}
MAX_LAYERS = 96


class LayerShape(C.Structure):
    _fields_ = [("outer", C.c_int64), ("C", C.c_int32), ("inner", C.c_int64), ("layout", C.c_int32)]


CONV_MAX_TAPS = 9


class WgradDesc(C.Structure):
    """vitta_wgrad_desc of include/vitta_hip.h (field for field)."""
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("grad_w", C.c_void_p), ("pro_bn", C.c_void_p * 4),
                ("pro_eps", C.c_float), ("src_off", C.c_void_p), ("src_mask", C.c_void_p),
                ("C", C.c_int32), ("K", C.c_int32), ("N", C.c_int32),
                ("Hs", C.c_int32), ("Ws", C.c_int32), ("Hg", C.c_int32), ("Wg", C.c_int32), ("sstride", C.c_int32),
                ("ntaps", C.c_int32), ("wtaps", C.c_int32),
                ("dh", C.c_int8 * 9), ("dw", C.c_int8 * 9), ("wt", C.c_int8 * 9), ("flags", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]
CONV_PRO_BN_RELU, CONV_EPI_APPLY, CONV_EPI_RELU, CONV_STATS = 1, 2, 4, 8
CONV_RES, CONV_RES_HALF, CONV_BWD_BN, CONV_BWD_RELU, CONV_PARITY4, CONV_STATS_RAW, CONV_INJ_RAW, CONV_POOL = 16, 32, 64, 128, 256, 512, 1024, 2048
BN_BWD_INJ_RAW = 2
WGRAD_DEFER_REDUCE = 1024
LN_BRANCH_BF16, LN_Y_BF16, LN_GY_BF16, LN_GBRANCH_BF16 = 1, 2, 4, 8
CONV_KERNEL_TILE, CONV_KERNEL_SK, CONV_KERNEL_PW, CONV_KERNEL_B3 = 0, 1, 2, 3


class ConvDesc(C.Structure):
    """vitta_conv_desc of include/vitta_hip.h (field for field)."""
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("y_raw", C.c_void_p), ("res", C.c_void_p),
                ("pro_bn", C.c_void_p * 4), ("epi_bn", C.c_void_p * 4), ("bwd_bn", C.c_void_p * 4),
                ("pro_eps", C.c_float), ("epi_eps", C.c_float), ("bwd_eps", C.c_float),
                ("st_shift", C.c_void_p), ("st_s1", C.c_void_p), ("st_s2", C.c_void_p),
                ("bwd_x", C.c_void_p), ("bwd_mask", C.c_void_p),
                ("inj_mu", C.c_void_p), ("inj_a", C.c_void_p), ("inj_b", C.c_void_p), ("inj_gscale", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("C", C.c_int32), ("K", C.c_int32), ("N", C.c_int32),
                ("Hs", C.c_int32), ("Ws", C.c_int32), ("Hg", C.c_int32), ("Wg", C.c_int32), ("Hy", C.c_int32), ("Wy", C.c_int32),
                ("sstride", C.c_int32), ("ostride", C.c_int32), ("oa", C.c_int32), ("ob", C.c_int32),
                ("ntaps", C.c_int32),
                ("dh", C.c_int8 * CONV_MAX_TAPS), ("dw", C.c_int8 * CONV_MAX_TAPS), ("wt", C.c_int8 * CONV_MAX_TAPS),
                ("flags", C.c_int32), ("tile", C.c_int32),
                ("ksplit", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("w_b3", C.c_void_p),
                ("cls_ntaps", C.c_int8 * 4),
                ("pool", C.c_void_p), ("pool_scale", C.c_float)]


class ColsumItem(C.Structure):
    """vitta_colsum_item of include/vitta_hip.h (field for field)."""
    _fields_ = [("d_partial", C.c_void_p), ("n_partials", C.c_int64), ("C", C.c_int32), ("cnt_value", C.c_float),
                ("d_out_a", C.c_void_p), ("d_out_b", C.c_void_p), ("d_cnt", C.c_void_p)]


_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/vitta_hip.h
SIGNATURES = {
    "vitta_abi_version": (C.c_int, []),
    "vitta_status_string": (C.c_char_p, [C.c_int]),
    "vitta_plan_create": (C.c_int, [C.POINTER(LayerShape), C.c_int, C.c_int, C.POINTER(_p)]),
    "vitta_plan_create_split": (C.c_int, [C.POINTER(LayerShape), C.c_int, C.c_int, C.POINTER(_i32), C.POINTER(_p)]),
    "vitta_plan_set_option": (C.c_int, [_p, C.c_int, C.c_int]),
    "vitta_plan_table_bytes": (_sz, [_p]),
    "vitta_plan_upload": (C.c_int, [_p, _p, _sz, _p]),
    "vitta_plan_destroy": (None, [_p]),
    "vitta_plan_total_channels": (_i64, [_p]),
    "vitta_plan_channel_offset": (_i64, [_p, C.c_int]),
    "vitta_plan_workspace_bytes": (_sz, [_p]),
    "vitta_plan_num_blocks": (_i64, [_p]),
    "vitta_moments_batched_f32": (C.c_int, [_p, C.POINTER(_p), _p, _p, _p, _p, _p, _sz, _p]),
    "vitta_moments_partials_f32": (C.c_int, [_p, C.POINTER(_p), _p, _sz, _p]),
    "vitta_moments_batched_bf16": (C.c_int, [_p, C.POINTER(_p), _p, _p, _p, _p, _p, _sz, _p]),
    "vitta_moments_partials_bf16": (C.c_int, [_p, C.POINTER(_p), _p, _sz, _p]),
    "vitta_moments_finalize_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _sz, _p]),
    "vitta_moments_to_meanvar_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "vitta_moments_workspace_bytes": (_sz, [_i64, _i32, _i64, _i32]),
    "vitta_moments_nchw_f32": (C.c_int, [_p, _i64, _i32, _i64, _p, _p, _p, _sz, _p]),
    "vitta_moments_nhwc_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _sz, _p]),
    "vitta_moments_nchw_bf16": (C.c_int, [_p, _i64, _i32, _i64, _p, _p, _p, _sz, _p]),
    "vitta_moments_nhwc_bf16": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _sz, _p]),
    "vitta_stat_align_fwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _f32, C.c_int,
                                           _p, _p, _p, _p, _p, _p, _p]),
    "vitta_stat_align_bwd_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _i32, _p, _p, _p, _p, _p]),
    "vitta_pred_consis_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _p, _p]),
    "vitta_tam_pool_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "vitta_tam_agg_fwd_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "vitta_tam_agg_bwd_f32": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    "vitta_tam_pool_bwd_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "vitta_tam_branch_supported": (C.c_int, [_i32, _i32]),
    "vitta_tam_branch_fused_supported": (C.c_int, [_i32, _i32, _i32]),
    "vitta_tam_branch_wgrad_f32": (C.c_int, [_p, _i32, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    "vitta_tam_branch_fwd_f32": (C.c_int, [_p, _p, C.POINTER(_p), _f32, _p, _p, C.POINTER(_p), _f32, _p, _i32, _i32, _i32,
                                           _p, _p, _p, _i32, _p]),
    "vitta_tam_branch_bwd_f32": (C.c_int, [_p, _p, C.POINTER(_p), _f32, _p, _p, C.POINTER(_p), _f32, _p, _i32, _i32, _i32,
                                           _p, _p, _p, _p, _p, _p, C.POINTER(_p), C.POINTER(_p), _i32, _p]),
    "vitta_tam_branch_fwd_fused_f32": (C.c_int, [_p, _p, C.POINTER(_p), _f32, _p, _p, C.POINTER(_p), _f32, _p, _i32, _i32, _i32,
                                                 _p, _p, _p, _p, _i32, _p]),
    "vitta_tam_branch_bwd_fused_f32": (C.c_int, [_p, _p, C.POINTER(_p), _f32, _p, _p, C.POINTER(_p), _f32, _p, _i32, _i32, _i32,
                                                 _p, _p, _p, _p, _p, _p, C.POINTER(_p), C.POINTER(_p), _p, _i32, _p]),
    "vitta_bn_act_partial_floats": (_sz, [_i64, _i32, _i64, _i32]),
    "vitta_bn_act_fwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _f32, _i64, _i32, _i64, _i32, _i32, _p, _p]),
    "vitta_bn_act_bwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f32, _p, _p, _p, _p, _i64, _i32, _i64, _i32,
                                       _i32, _p, _p, _p, _i32, _p]),
    "vitta_plan_layer_geometry": (C.c_int, [_p, C.c_int, C.POINTER(_i64)]),
    "vitta_wmsa_supported": (C.c_int, [_i32, _i32]),
    "vitta_wmsa_rel_supported": (C.c_int, [_i32, _i32]),
    "vitta_wmsa_fwd_f32": (C.c_int, [_p, _p, _p, _i32, _i64, _i32, _i32, _i32, _f32, _p, _p, _p]),
    "vitta_wmsa_rel_fwd_f32": (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i32, _i64, _i32, _i32, _i32, _f32, _p, _i32, _i64,
                                         _p, _p, _p]),
    "vitta_wmsa_rel_bwd_f32": (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i32, _i64, _i32, _i32, _i32, _f32, _p, _i32, _i64,
                                         _p, _p, _p, _p, _p, _p, _p]),
    "vitta_wmsa_bf16_supported": (C.c_int, [_i32, _i32, _i32]),
    "vitta_wmsa_rel_fwd_bf16": (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i32, _i64, _i32, _i32, _i32, _f32, _p, _i32, _i64,
                                          _p, _p, _p]),
    "vitta_wmsa_rel_bwd_bf16": (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i32, _i64, _i32, _i32, _i32, _f32, _p, _i32, _i64,
                                          _p, _p, _p, _p, _p, _p]),
    "vitta_wmsa_rel_fwd_bf16_io": (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i32, _i64, _i32, _i32, _i32, _f32, _p, _i32, _i64, _p, _p, _i32, _p]),
    "vitta_wmsa_rel_bwd_bf16_io": (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i32, _i64, _i32, _i32, _i32, _f32, _p, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "vitta_wmsa_bf16_dtable_workspace_bytes": (C.c_size_t, [_i64, _i32, _i32]),
    "vitta_gemm_tn_bf16_supported": (C.c_int, [_i64, _i32, _i32]),
    "vitta_gemm_tn_bf16_workspace_bytes": (C.c_size_t, [_i64, _i32, _i32]),
    "vitta_gemm_tn_bf16": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "vitta_wmsa_bf16_dtable_supported": (C.c_int, [_i32, _i32, _i32]),
    "vitta_wmsa_bwd_f32": (C.c_int, [_p, _p, _p, _i32, _i64, _i32, _i32, _i32, _f32, _p, _p, _p, _p, _p, _p, _p]),
    "vitta_moments_partials_timed_f32": (C.c_int, [_p, C.POINTER(_p), _p, _sz, _p, _p, _p]),
    "vitta_event_create": (C.c_int, [C.POINTER(_p)]),
    "vitta_event_destroy": (None, [_p]),
    "vitta_event_elapsed_ms": (C.c_int, [_p, _p, C.POINTER(_f32)]),
    "vitta_conv_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "vitta_conv_num_blocks": (_i64, [C.POINTER(ConvDesc)]),
    "vitta_conv_f32": (C.c_int, [C.POINTER(ConvDesc), _p]),
    "vitta_conv_workspace_bytes": (_sz, [C.POINTER(ConvDesc)]),
    "vitta_conv_wgrad_f32": (C.c_int, [C.POINTER(WgradDesc), _p]),
    "vitta_gemm_bf16x_supported": (C.c_int, [_i64, _i64, _i64]),
    "vitta_gemm_nt_bf16x_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "vitta_gemm_nt_bf16x": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _p]),
    "vitta_conv_wgrad_reduce_f32": (C.c_int, [C.POINTER(C.POINTER(WgradDesc)), _i32, _p]),
    "vitta_conv_repack_f32": (C.c_int, [_p, _i32, _i64, _p]),
    "vitta_conv_pack_b3_bytes": (_sz, [_i32, _i32, _i32]),
    "vitta_conv_pack_b3": (C.c_int, [_p, _p, _i32, _i32, _i32, _p]),
    "vitta_conv_pack_b3_table": (C.c_int, [_p, _i32, _i64, _p]),
    "vitta_conv_timed_f32": (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p]),
    "vitta_conv_flops": (_i64, [C.POINTER(ConvDesc)]),
    "vitta_conv_fastdiv_host": (_i64, [_i64, _i64]),
    "vitta_conv_kernel": (C.c_int, [C.POINTER(ConvDesc)]),
    "vitta_stem_conv7_f32": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p]),
    "vitta_stem_conv7_wgrad_workspace_bytes": (_sz, []),
    "vitta_stem_conv7_wgrad_f32": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _sz, _p]),
    "vitta_stem_bn_relu_pool_bwd_f32": (C.c_int, [_p, _p, C.POINTER(_p), _f32, _i64, _i32, _i32, _i32, _p, _p, _p, _p]),
    "vitta_linear_fwd_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _p, _p]),
    "vitta_linear_bwd_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p, _p]),
    "vitta_tanet_head_lds_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "vitta_tanet_head_fwd_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "vitta_tanet_head_bwd_f32": (C.c_int, [_p, _p, _p, _p, _p, C.c_float, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    "vitta_loss_axpby_f32": (C.c_int, [_p, _p, C.c_float, C.c_float, _p, _p, _p, _p]),
    "vitta_loss_axpby_bwd_f32": (C.c_int, [_p, C.c_float, C.c_float, _p, _p, _p]),
    "vitta_gemm_nt_supported": (C.c_int, [_i64, _i32, _i32]),
    "vitta_gemm_nt_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "vitta_gemm_nt_sk_workspace_bytes": (C.c_int64, [_i32]),
    "vitta_gemm_nt_sk_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "vitta_gemm_nt_bf16w_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "vitta_tam_pool_cm_f32": (C.c_int, [_p, C.POINTER(_p), _f32, _i32, _i32, _i32, _i32, _p, _p]),
    "vitta_tam_agg_fwd_cm_f32": (C.c_int, [_p, C.POINTER(_p), _f32, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "vitta_tam_agg_bwd_cm_f32": (C.c_int, [_p, C.POINTER(_p), _f32, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    "vitta_bn_bwd_cm_f32": (C.c_int, [_p, _p, _p, _p, _p, _f32, C.POINTER(_p), _f32, _p, _p, _p, _p, _i32, _p, _p, _p, _p,
                                      _i32, _i32, _i32, _i32, _p]),
    "vitta_avgpool_cm_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "vitta_avgpool_cm_bwd_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "vitta_ln_supported": (C.c_int, [_i32]),
    "vitta_ln_num_partials": (_i64, [_i64]),
    "vitta_ln_fwd_f32": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p, _f32, _p, _p, _p, _p, _p, _p, _p]),
    "vitta_ln_bwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p]),
    "vitta_ln_fwd_mixed": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p, _f32, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "vitta_ln_bwd_mixed": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _i32, _p]),
    "vitta_colsum2_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _f32, _p]),
    "vitta_colsum2_multi_f32": (C.c_int, [C.POINTER(ColsumItem), _i32, _p]),
    "vitta_stem_bn_relu_pool_fwd_f32": (C.c_int, [_p, C.POINTER(_p), _f32, _i64, _i32, _i32, _i32, _p, _p]),
    "vitta_stem_bn_relu_pool_bwd_affine_f32": (C.c_int, [_p, _p, C.POINTER(_p), _f32, _i64, _i32, _i32, _i32, _p, _p, _p]),
    "vitta_stem_bn_relu_pool_fwd_cm_f32": (C.c_int, [_p, C.POINTER(_p), _f32, _i64, _i32, _i32, _i32, _p, _p]),
    "vitta_stem_bn_relu_pool_bwd_cm_f32": (C.c_int, [_p, _p, C.POINTER(_p), _f32, _i64, _i32, _i32, _i32, _p, _p, _p, _p]),
    "vitta_frames_resample_norm_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _p, _i32, _p, _p, _i32,
                                                 _i32, _i32, _i32, _p]),
    "vitta_frames_cv2_resize": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i32, _i32, _p, _p, _i32, _p, _p, _p]),
    "vitta_scale_add_f32": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "vitta_patch_gather_f32": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "vitta_adam_step_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _f32, _f32, _f32, _f32, _f32, _i64, _p]),
    "vitta_sgd_step_f32": (C.c_int, [_p, _p, _p, _f32, _f32, _f32, _i64, _p]),
}

_LIB = None


class VittaHipError(RuntimeError):
    pass


_SAID = set()


def loud_once(key, msg):
    """A vendor-library / ATen path is about to run where a hand-written kernel exists (an A/B switch, or a shape a kernel declines): say
    so ONCE per cause on stderr and through logging -- tests guard the default, this guards a production run."""
    if key in _SAID:
        return
    _SAID.add(key)
    import logging
    import sys
    text = "[vitta_amd] WARNING: " + msg
    logging.getLogger("vitta_amd").warning(text)
    print(text, file=sys.stderr, flush=True)


def lib():
    """Load (once) and return the bound library; raise loudly if it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise VittaHipError(
            f"{LIB_PATH} not found: build it with `python -m vitta_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
    # Load order matters: PyTorch-ROCm ships its own libamdhip64; importing torch (and creating its HIP
    # context) first makes libvitta_hip.so bind to the SAME runtime instance that owns the tensors and
    # streams it is handed.  Loaded the other way round, launches on torch's streams fail.
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise VittaHipError(f"libvitta_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _LIB = handle
    return _LIB


# {ABI entry point name: calls} while it is a dict (tests assert which path ran; the product leaves it None)
CALL_COUNTS = None


def check(status, what=""):
    if CALL_COUNTS is not None:
        CALL_COUNTS[what] = CALL_COUNTS.get(what, 0) + 1
    if status != 0:
        msg = lib().vitta_status_string(status).decode()
        raise VittaHipError(f"{what}: {msg} (status {status})")
